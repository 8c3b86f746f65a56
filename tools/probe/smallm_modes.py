"""Small codebooks (M = 16, 128): kernel ms of the LDS-table pass (mode 1) against assignment only (mode 0)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
P, S = 36, 1 << 21
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
for M in (16, 128):
    s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, M // 2); s.grow()
    s.enable_timing(True)
    for mode in ("1", "0", "3"):
        os.environ["ECOZ2_VQ_FORCE_MODE"] = mode
        ts = []
        for i in range(5):
            s.run_pass(); ts.append(s.last_pass_kernel_ms())
        print("M", M, "mode", mode, "kernel ms", np.round(ts, 3), " GB/s of frames", round(S * 296 / min(ts) * 1e-6))
    del os.environ["ECOZ2_VQ_FORCE_MODE"]
    s.close()
