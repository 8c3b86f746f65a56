#!/bin/bash
# round 4, final build (recorded accumulate): the whole GPU suite, smoke(), the fuzz modes, the bench line and its
# collective variants, the file entry points at BASELINE sizes
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/rf2_tests.log 2>&1; rc=$?
tail -4 gpurun_out/rf2_tests.log
grep -q "Memory access fault" gpurun_out/rf2_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/rf2_smoke.txt 2>&1 || { tail -5 gpurun_out/rf2_smoke.txt; exit 1; }
tail -1 gpurun_out/rf2_smoke.txt
{ timeout -k 10 300 python tools/fuzz_parity.py 400 9401; timeout -k 10 500 python tools/fuzz_parity.py 400 9402 pre; timeout -k 10 200 python tools/fuzz_parity.py 100 9403 hmm; } > gpurun_out/rf2_fuzz.txt 2>&1 || { tail -5 gpurun_out/rf2_fuzz.txt; exit 1; }
grep "done" gpurun_out/rf2_fuzz.txt
python bench.py > gpurun_out/rf2_bench.json 2> gpurun_out/rf2_bench.err || { tail -5 gpurun_out/rf2_bench.err; exit 1; }
python bench.py --force-collective --no-cpu-baseline > gpurun_out/rf2_bench_fc.json 2> gpurun_out/rf2_bench_fc.err || exit 1
python bench.py --gpus 2 --in-process --no-cpu-baseline > gpurun_out/rf2_bench_inproc2.json 2> gpurun_out/rf2_bench_inproc2.err || exit 1
python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-extras > gpurun_out/rf2_bench_gloo2.json 2> gpurun_out/rf2_bench_gloo2.err || exit 1
python tools/probe/scale_check.py > gpurun_out/rf2_scale_check.txt 2>&1 || { tail -5 gpurun_out/rf2_scale_check.txt; exit 1; }
tail -3 gpurun_out/rf2_scale_check.txt
python3 - <<'PY'
import json
for f in ['rf2_bench','rf2_bench_fc','rf2_bench_inproc2','rf2_bench_gloo2']:
    d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, "value %.4f G step %.4f kernel %.4f acc %.4f frac %.3f parity %s" % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('accumulate_kernel_ms',0), d['roofline']['frac'], d['config']['parity']['ok']), (d['config'].get('collective') or {}).get('allreduce_us_per_call'))
PY
