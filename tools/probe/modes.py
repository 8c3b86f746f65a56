import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
P, M, S = 36, 1024, 1 << 21
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, 512); s.grow()
s.enable_timing(True)
sym = torch.empty(S, dtype=torch.int16, device="cuda"); dmin = torch.empty(S, dtype=torch.float64, device="cuda")
for mode in ("2", "3", "0", "2", "3"):
    os.environ["ECOZ2_VQ_FORCE_MODE"] = mode
    ts = []
    for i in range(6):
        s.run_pass(sym, dmin); ts.append(s.last_pass_kernel_ms())
    print("mode", mode, "kernel ms", np.round(ts, 3))
for mode, outs in (("0", False), ("2", False)):
    os.environ["ECOZ2_VQ_FORCE_MODE"] = mode
    ts = []
    for i in range(6):
        s.run_pass(None, None); ts.append(s.last_pass_kernel_ms())
    print("mode", mode, "no outputs: kernel ms", np.round(ts, 3))
