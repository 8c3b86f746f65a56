#!/bin/bash
# the recorded accumulate as the default: the whole GPU suite, the pre fuzz, then bench.py with and without it (same box)
R=$GRAFT_REPO_ROOT
cd $R
export ECOZ2_VQ_QUIET=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/rec7_tests.log 2>&1; rc=$?
tail -6 gpurun_out/rec7_tests.log
grep -q "Memory access fault" gpurun_out/rec7_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/fuzz_parity.py 200 9102 pre > gpurun_out/rec7_fuzz.txt 2>&1 || { tail -5 gpurun_out/rec7_fuzz.txt; exit 1; }
tail -2 gpurun_out/rec7_fuzz.txt
for r in 1 0 1 0; do
  ECOZ2_VQ_RECORDS=$r python bench.py --no-cpu-baseline --no-extras > gpurun_out/rec7_bench_$r.json 2> gpurun_out/rec7_bench_$r.err || { tail -5 gpurun_out/rec7_bench_$r.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/rec7_bench_$r.json").read().strip().splitlines()[-1])
print("RECORDS=$r value %.4f G  ms_per_step %.4f  kernel_ms %.4f  accumulate_kernel_ms %.4f parity %s" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"].get("accumulate_kernel_ms",0), d["config"]["parity"]["ok"]))
PY
done
