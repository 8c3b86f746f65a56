import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
import numpy as np
import ecoz2rs_amd as e
import tempfile
tmp = tempfile.mkdtemp()
frames = e.synth.synth_frames(20241, 4, 36, 0, 10000)
f = os.path.join(tmp, "data", "predictors", "_", "all.prd")
e.formats.write_prd(f, "_", frames)
os.environ["ECOZ2_VQ_OUT_ROOT"] = tmp
os.environ["ECOZ2_VQ_MAX_CODEBOOK_SIZE"] = "16"
os.environ["ECOZ2_VQ_GPUS"] = "1"
os.environ["ECOZ2_VQ_COLLECTIVE"] = "rccl"
os.environ["NCCL_DEBUG"] = "INFO"
try:
    e.vq_learn(None, 36, 0.05, "_", [f])
    print("OK")
except Exception as ex:
    print("FAILED", ex)
