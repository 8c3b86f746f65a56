#!/bin/bash
# round 5: the bench line (with the 16 M-frame extra, the small corpus and the CPU baseline), its collective variants, the file
# entry points at BASELINE sizes
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
python bench.py > gpurun_out/rj_bench.json 2> gpurun_out/rj_bench.err || { tail -5 gpurun_out/rj_bench.err; exit 1; }
python bench.py --force-collective --no-cpu-baseline --no-extras > gpurun_out/rj_bench_fc.json 2> gpurun_out/rj_bench_fc.err || exit 1
ECOZ2_BENCH_SKIP_16M=1 ECOZ2_BENCH_SKIP_SMALL=1 python bench.py --gpus 2 --in-process --no-cpu-baseline > gpurun_out/rj_bench_inproc2.json 2> gpurun_out/rj_bench_inproc2.err || exit 1
python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-extras > gpurun_out/rj_bench_gloo2.json 2> gpurun_out/rj_bench_gloo2.err || exit 1
python tools/probe/scale_check.py > gpurun_out/rj_scale_check.txt 2>&1 || { tail -5 gpurun_out/rj_scale_check.txt; exit 1; }
tail -3 gpurun_out/rj_scale_check.txt
python3 - <<'PY'
import json
for f in ['rj_bench','rj_bench_fc','rj_bench_inproc2','rj_bench_gloo2']:
    d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    c=d['config']
    print(f, "value %.4f G step %.4f kernel %.4f frac %.3f parity %s" % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], (c.get('parity') or {}).get('ok')), (c.get('collective') or {}).get('allreduce_us_per_call'))
    e=c.get('learn_end_to_end')
    if e:
        print('  e2e %.2f ms' % (1e3*e['seconds']))
        for l in e['levels']: print('   M %5d passes %d kernel %.3f step %.3f frac %.2f ar %s' % (l['M'], l['passes'], l['kernel_ms'], l['step_ms'], l['frac_of_bound'], l.get('allreduce_us_per_call')))
    if c.get('strong_scaling_16M'): print('  16M:', {k: c['strong_scaling_16M'].get(k) for k in ('level_ms_per_pass','level_kernel_ms_per_pass','ladder_seconds','level_frames_per_sec')})
    if c.get('small_corpus'): print('  small:', {k: v for k, v in c['small_corpus'].items() if k != 'levels'})
    if 'cpu_baseline' in d: print('  cpu:', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
