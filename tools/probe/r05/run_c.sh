#!/bin/bash
# round 5: first run of the split pass (sort + candidate sweep + finish + reduce): parity first, then the bench A/B
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/rc_smoke.txt 2>&1; rc=$?; tail -3 gpurun_out/rc_smoke.txt
grep -q "Memory access fault" gpurun_out/rc_smoke.txt && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/rc_tests.log 2>&1; rc=$?
tail -5 gpurun_out/rc_tests.log
grep -q "Memory access fault" gpurun_out/rc_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
export ECOZ2_BENCH_SKIP_16M=1 ECOZ2_BENCH_SKIP_SMALL=1
for v in new old; do
  if [ $v = old ]; then export ECOZ2_VQ_SPLIT_SWEEP=0; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 21 > gpurun_out/rc_bench_$v.json 2> gpurun_out/rc_bench_$v.err || { tail -5 gpurun_out/rc_bench_$v.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/rc_bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'G %.3f step %.3f kernel %.3f acc %.3f parity %s e2e %.2f ms' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('accumulate_kernel_ms',0), d['config']['parity']['ok'], 1e3*d['config']['learn_end_to_end']['seconds']))
for l in d['config']['learn_end_to_end']['levels']: print('   M %5d passes %d kernel %.3f step %.3f' % (l['M'], l['passes'], l['kernel_ms'], l['step_ms']))
PY
done
