import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
from tests import oracle_lib
o = oracle_lib.load()
P = 36
os.environ["ECOZ2_VQ_QUIET"] = "1"
os.environ["ECOZ2_VQ_PREFILTER_MIN_M"] = "64"
frames = e.synth.synth_frames(5, 4, P, 0, 6000)
with e.VqSession(P) as s:
    s.set_frames(frames); s.prepare(); s.init_codebook()
    lv = s.learn(0.05, 128)
    print("ok", [l.passes for l in lv], lv[-1].DD, flush=True)
rc, lo, _ = o.learn(frames, 0.05, 128)
print("oracle DD", lo[-1]["DD"], "match", lo[-1]["DD"] == lv[-1].DD)
