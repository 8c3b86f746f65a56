export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_ROBUSTNESS=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for rep in 1 2; do
for v in head base; do
  if [ "$v" = base ]; then unset ECOZ2VQ_LIB; else export ECOZ2VQ_LIB=$PWD/tools/probe/ab/$v/libecoz2vq.so; fi
  for m in 1024 256; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --codebook-size $m > gpurun_out/ab_${v}_${m}_$rep.json 2> gpurun_out/ab_${v}_${m}_$rep.err || { tail -5 gpurun_out/ab_${v}_${m}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_${v}_${m}_$rep.json") if l.startswith("{")][-1])
a=d["config"].get("weak_scaling_anchor") or {}
print("$v M=$m rep $rep: value %.3f G kernel %.4f ms step %.4f | anchor kernel %s" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["ms_per_step"], a.get("kernel_ms")))
PY
  done
done
done
