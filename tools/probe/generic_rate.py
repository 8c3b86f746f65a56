import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
os.environ["ECOZ2_VQ_QUIET"] = "1"
for P in (33, 36, 35, 38, 10, 4):
    T, M = 1 << 20, 1024
    frames = e.synth.synth_frames(1, 20, P, 0, T)
    s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(1e9, M); s.enable_timing(True)
    ks = []
    for i in range(3):
        s.run_pass(); ks.append(s.last_pass_kernel_ms())
    k = min(ks)
    print(f"P={P}: M={M} pass on {T} frames {k:.3f} ms = {T/k*1e-6:.3f} G frames/s, {2*M*(P+1)*T/k*1e-9:.1f} TF useful")
    s.close()
