"""Ablations of the prefiltered pass at M = 1024 (ECOZ2VQ_LIB picks the library variant):
kernel ms with accumulate (mode 2) / assignment only (mode 0), with and without outputs."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
P, S = 36, 1 << 21
M = int(os.environ.get("PROBE_M", "1024"))
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, M // 2); s.grow()
s.enable_timing(True)
sym = torch.empty(S, dtype=torch.int16, device="cuda"); dmin = torch.empty(S, dtype=torch.float64, device="cuda")
for mode in ("2", "0", "2", "0"):
    os.environ["ECOZ2_VQ_FORCE_MODE"] = mode
    ts = []
    for i in range(6):
        s.run_pass(sym, dmin); ts.append(s.last_pass_kernel_ms())
    print("lib", os.environ.get("ECOZ2VQ_LIB", "default"), "mode", mode, "kernel ms", np.round(ts, 3), "info", s.last_pass_info())
