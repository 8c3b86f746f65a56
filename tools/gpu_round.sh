#!/bin/bash
# One parameterised driver for the GPU-box steps of a round (replaces the per-call scripts of rounds 4 and 5).
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh <step>...'      every step writes gpurun_out/<tag>_<step>.*
# steps:  tests | smoke | fuzz[:N[:seed[:pre|gen|hmm]]] | bench[:extra args] | flagged | fallback | scale | stamps | exp[:modes] | profiles[:tag[:bench args]]
#         | quantprof[:tag] | ab:<variant>... (variants built by tools/probe/ab/build_variant.sh; "base" = the product)
# TAG (environment, default r06) prefixes the output files.  A failing step ends the call (no GPU step is started behind it).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export ECOZ2_VQ_QUIET=1
TAG=${TAG:-r06}
OUT=gpurun_out
mkdir -p $OUT
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "== $step"
  case $name in
    tests)   timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_tests.log 2>&1 || { tail -30 $OUT/${TAG}_tests.log; exit 1; }
             tail -3 $OUT/${TAG}_tests.log ;;
    smoke)   timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/${TAG}_smoke.txt 2>&1 || { tail -20 $OUT/${TAG}_smoke.txt; exit 1; }
             tail -1 $OUT/${TAG}_smoke.txt ;;
    fuzz)    IFS=: read -r n seed mode <<< "$arg"
             mode=${mode:-pre}; marg=$mode; [ "$mode" = gen ] && marg=""   # (gen: the general mode -- random T, M, P; one pass + update + quantize)
             timeout -k 10 1100 python tools/fuzz_parity.py ${n:-300} ${seed:-9601} $marg > $OUT/${TAG}_fuzz_${mode}.txt 2>&1 || { tail -20 $OUT/${TAG}_fuzz_${mode}.txt; exit 1; }
             tail -2 $OUT/${TAG}_fuzz_${mode}.txt ;;
    bench)   timeout -k 10 900 python bench.py $arg > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -20 $OUT/${TAG}_bench.err; exit 1; }
             python tools/bench_digest.py "python bench.py $arg"=$OUT/${TAG}_bench.json | cut -c1-900 ;;
    flagged) timeout -k 10 600 python tools/probe/flagged_probe.py $arg > $OUT/${TAG}_flagged.txt 2>&1 || { tail -20 $OUT/${TAG}_flagged.txt; exit 1; }
             cut -c1-400 $OUT/${TAG}_flagged.txt ;;
    fallback) timeout -k 10 600 python tools/probe/fallback_probe.py $arg > $OUT/${TAG}_fallback.txt 2>&1 || { tail -20 $OUT/${TAG}_fallback.txt; exit 1; }
             cut -c1-400 $OUT/${TAG}_fallback.txt ;;
    scale)   timeout -k 10 900 python tools/probe/scale_check.py $arg > $OUT/${TAG}_scale_check.txt 2>&1 || { tail -20 $OUT/${TAG}_scale_check.txt; exit 1; }
             cut -c1-300 $OUT/${TAG}_scale_check.txt ;;
    stamps)  ECOZ2VQ_LIB=tools/probe/ab/stamp/libecoz2vq.so timeout -k 10 400 python tools/probe/sweep_stamps.py > $OUT/${TAG}_stamps.txt 2>&1 || { tail -20 $OUT/${TAG}_stamps.txt; exit 1; }
             grep -v "pass 1" $OUT/${TAG}_stamps.txt | cut -c1-420 ;;
    exp)     EXP_MODES=${arg:-0,1,2,3,0} ECOZ2VQ_LIB=tools/probe/ab/stamp/libecoz2vq.so timeout -k 10 400 python tools/probe/sweep_exp.py > $OUT/${TAG}_exp.txt 2>&1 || { tail -20 $OUT/${TAG}_exp.txt; exit 1; }
             cut -c1-330 $OUT/${TAG}_exp.txt ;;
    profiles) IFS=: read -r ptag pargs <<< "$arg"
             ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_ROBUSTNESS=1 bash tools/profile_bench.sh ${ptag:-$TAG} $pargs || exit 1 ;;
    quantprof) QT=${arg:-r06q}; ( cd /tmp && export TMPDIR=/tmp && R=$OLDPWD && Q="python3 $R/tools/quantize_profile.py" &&
               rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/${QT}_kt" -- $Q > "$R/$OUT/${QT}_kt.json" 2> "$R/$OUT/${QT}_kt.err" && echo "kernel trace done" &&
               rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/${QT}_fetch" -- $Q 10000000 3 > "$R/$OUT/${QT}_fetch.json" 2> "$R/$OUT/${QT}_fetch.err" && echo "fetch done" &&
               rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/${QT}_write" -- $Q 10000000 3 > "$R/$OUT/${QT}_write.json" 2> "$R/$OUT/${QT}_write.err" && echo "write done" &&
               rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$R/$OUT/${QT}_sq1" -- $Q 10000000 3 > "$R/$OUT/${QT}_sq1.json" 2> "$R/$OUT/${QT}_sq1.err" && echo "sq1 done" &&
               rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d "$R/$OUT/${QT}_sq2" -- $Q 10000000 3 > "$R/$OUT/${QT}_sq2.json" 2> "$R/$OUT/${QT}_sq2.err" && echo "sq2 done" ) || { tail -5 $OUT/${QT}_*.err; exit 1; } ;;
    ab)      for v in ${arg//:/ }; do
               if [ "$v" = base ]; then unset ECOZ2VQ_LIB; else export ECOZ2VQ_LIB=$PWD/tools/probe/ab/$v/libecoz2vq.so; fi
               ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_ROBUSTNESS=1 timeout -k 10 600 python bench.py --no-cpu-baseline > $OUT/${TAG}_ab_$v.json 2> $OUT/${TAG}_ab_$v.err || { tail -20 $OUT/${TAG}_ab_$v.err; exit 1; }
               python tools/bench_digest.py "$v"=$OUT/${TAG}_ab_$v.json | cut -c1-700
             done; unset ECOZ2VQ_LIB ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
