#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1 ECOZ2_BENCH_SKIP_16M=1 ECOZ2_BENCH_SKIP_SMALL=1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/rh_smoke.txt 2>&1; rc=$?; tail -2 gpurun_out/rh_smoke.txt
grep -q "Memory access fault" gpurun_out/rh_smoke.txt && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_prefilter.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/rh_tests.log 2>&1; rc=$?
tail -3 gpurun_out/rh_tests.log
grep -q "Memory access fault" gpurun_out/rh_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
for m in 1024 512 256 128; do
  X="--no-extras"; [ $m = 1024 ] && X=""
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 21 --codebook-size $m $X > gpurun_out/rh_bench_$m.json 2> gpurun_out/rh_bench_$m.err || { tail -5 gpurun_out/rh_bench_$m.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/rh_bench_$m.json').read().strip().splitlines()[-1])
print('M $m', 'G %.3f step %.3f sweep %.3f other pass kernels %.3f parity %s' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('accumulate_kernel_ms',0), d['config']['parity']['ok']))
e=d['config'].get('learn_end_to_end')
if e:
    print('  e2e %.2f ms' % (1e3*e['seconds']))
    for l in e['levels']: print('   M %5d passes %d kernel %.3f step %.3f' % (l['M'], l['passes'], l['kernel_ms'], l['step_ms']))
PY
done
true
