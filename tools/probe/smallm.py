import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
P, S = 36, 1 << 21
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.enable_timing(True)
while s.codebook_size() < 128:
    s.grow(); M = s.codebook_size()
    out = []
    for mode in (None, "0", "3"):
        if mode is None: os.environ.pop("ECOZ2_VQ_FORCE_MODE", None)
        else: os.environ["ECOZ2_VQ_FORCE_MODE"] = mode
        ks = []
        for it in range(4):
            s.run_pass(); ks.append(s.last_pass_kernel_ms())
        out.append(min(ks))
    os.environ.pop("ECOZ2_VQ_FORCE_MODE", None)
    s.run_pass(); s.pass_stats(); s.update()
    print(f"M={M:4d}  LDS-table pass {out[0]:.3f} ms | sweep only {out[1]:.3f} ms | sweep+images, no atomics (global-mode code) {out[2]:.3f} ms")
