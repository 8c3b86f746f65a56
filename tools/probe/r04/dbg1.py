"""one M = 1024 pass through the accumulating prefiltered kernel, checked against the oracle (debug aid)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
from tests import oracle_lib
os.environ["ECOZ2_VQ_QUIET"] = "1"
os.environ["ECOZ2_VQ_PREFILTER_MIN_M"] = "64"
os.environ["ECOZ2_VQ_PLAIN_FIRST"] = "0"
P = 36
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
oracle = oracle_lib.load()
frames = e.synth.synth_frames(20250, 20, P, 0, T)
rc, levels, _ = oracle.learn(frames[:4096], 0.5, M)
refl = levels[-1]["reflections"]
cq = oracle.reflections_to_cq(refl)
sym_o, dmin_o = oracle.quantize(cq, frames)
with e.VqSession(P) as s:
    s.set_frames(frames); s.prepare(); s.set_codebook(refl)
    s.run_pass()
    print("pass launched", flush=True)
    s.synchronize()
    print("synchronized", flush=True)
    rows = s.get_rows()
    used, nfb = s.last_pass_info()
    sym, dmin = s.quantize(frames)
print("prefiltered", used, "fallback", nfb, "counts ok", int(rows.reshape(M, -1)[:, 74].sum()) == T,
      "quantize equal", np.array_equal(sym, sym_o), np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64)))
cnt_o = np.bincount(sym_o, minlength=M)
print("cell counts equal the oracle's:", np.array_equal(rows.reshape(M, -1)[:, 74], cnt_o))
