"""Cycles of a k_pass_mfma wave by phase at the small levels (library built with -DE2VQ_MFMA_STAMP): real ladder, one line per pass."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
from ecoz2rs_amd._lib import lib
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
NAMES = ["frames", "sweep", "combine + prefetch", "accumulate", "final flush"]
frames = e.synth.synth_frames(20244, 20, P, 0, S)
out = (C.c_ulonglong * 16)()
with e.VqSession(P) as s:
    s.set_frames(frames); s.prepare(); s.init_codebook()
    s.enable_timing(True)
    for M in (2, 4, 8, 16, 32, 64, 128):
        s.grow()
        for p in range(2):
            lib.e2vq_debug_mfma_stamps(None, 1)
            s.run_pass(); s.synchronize()
            ms = s.last_pass_kernel_ms()
            lib.e2vq_debug_mfma_stamps(out, 0)
            v = np.array(out[:16], dtype=np.float64)
            nb, nw = max(v[8], 1), max(v[9], 1)
            print(f"M={M} pass {p}: kernel {ms:.3f} ms, cycles/block {v[:4].sum() / nb:.0f}: " +
                  "  ".join(f"{NAMES[k]}: {v[k] / nb:.0f}" for k in range(4)) + f"  | {NAMES[4]} per wave: {v[4] / nw:.0f}  "
                  f"({v[:5].sum() / nw / 1e3:.0f} kcyc per wave)", flush=True)
            s.pass_stats(); s.update()
