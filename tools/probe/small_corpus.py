"""The workload the reference documents (notes.md:122-153: 38 265 vectors, eps 0.05, M = 2 ... 2048) through this library:
prints bench.py's `config.small_corpus` section (GPU parts only) in readable form.  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/probe/small_corpus.py` the stats table tells which launch of a pass costs
what at this size (every pass is launch latency here)."""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch  # (first: its bundled HIP runtime initialises before the library's first HIP call)

assert torch.cuda.is_available()
torch.cuda.init()
import bench
import ecoz2rs_amd as e

d = bench.small_corpus(e, np, with_cpu=False)
lv = d.pop("levels", [])
print(json.dumps(d, indent=1))
for x in lv:
    print(f"  M = {x['M']:5d}: {x['passes']} passes, kernel {x['kernel_us_per_pass']:7.1f} us / pass, step {x['step_us_per_pass']:7.1f} us / pass")
