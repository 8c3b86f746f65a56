#!/bin/bash
# Same-box A/B of the product library against a variant built from the COMMITTED sources (tools/probe/ab/head/libecoz2vq.so: compile
# the changed translation units under `git stash`, link them with the product's other objects), on the bench's timed region, the
# M = 256 level, the quantize kernel and -- round 4's kernel -- the M = 128 level:  gpurun -- 'bash tools/probe/ab/run_head_ab.sh'
export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_SMALL=1 ECOZ2_BENCH_SKIP_ROBUSTNESS=1
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in head base; do
  if [ "$v" = base ]; then unset ECOZ2VQ_LIB; else export ECOZ2VQ_LIB=$PWD/tools/probe/ab/$v/libecoz2vq.so; fi
  for m in 1024 256 128; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --codebook-size $m > gpurun_out/ab_${v}_${m}_$rep.json 2> gpurun_out/ab_${v}_${m}_$rep.err || { tail -5 gpurun_out/ab_${v}_${m}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_${v}_${m}_$rep.json") if l.startswith("{")][-1])
a=d["config"].get("weak_scaling_anchor") or {}
q=d["config"].get("quantize") or {}
print("$v M=$m rep $rep: value %.3f G kernel %.4f ms step %.4f | anchor kernel %s | quantize ms %s" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["ms_per_step"], a.get("kernel_ms"), q.get("kernel_ms")))
PY
  done
  timeout -k 10 200 python tools/quantize_profile.py 10000000 6 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v quantize rep $rep: avg %.4f ms min %.4f' % (d['avg_ms'], min(d['event_ms'])))"
done
done
