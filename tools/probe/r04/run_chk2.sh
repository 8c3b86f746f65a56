#!/bin/bash
cd $GRAFT_REPO_ROOT
export ECOZ2_VQ_QUIET=1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/chk2_tests.log 2>&1; rc=$?
tail -4 gpurun_out/chk2_tests.log
grep -q "Memory access fault" gpurun_out/chk2_tests.log && exit 1
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/fuzz_parity.py 200 9502 pre > gpurun_out/chk2_fuzz.txt 2>&1 || { tail -5 gpurun_out/chk2_fuzz.txt; exit 1; }
tail -1 gpurun_out/chk2_fuzz.txt
for i in 1 2; do python bench.py --no-cpu-baseline > gpurun_out/chk_bench.json 2> gpurun_out/chk_bench.err || { tail -5 gpurun_out/chk_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/chk_bench.json").read().strip().splitlines()[-1])
le=d["config"]["learn_end_to_end"]
print("value %.4f G step %.4f kernel %.4f acc %.4f ladder %.2f ms parity %s" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["accumulate_kernel_ms"], le["seconds"]*1e3, d["config"]["parity"]["ok"]), [round(l["step_ms"],3) for l in le["levels"]])
PY
done
