"""Where do the cycles of a k_pass_pre wave go?  Run with ECOZ2VQ_LIB=tools/probe/ab/stamp1/libecoz2vq.so (library built
with -DE2VQ_PRE_STAMP=1: s_memtime stamps at the phase ends, vector-memory counter drained there; =2: stamps only).
Real ladder to M = 128 on 2^21 frames, then every pass of the levels 256 / 512 / 1024 with the real convergence rule."""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
from ecoz2rs_amd._lib import lib
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
NAMES = ["limb images (+ older atomics)", "tile loop", "certify + old cells", "FP64 frames", "gather 1", "exact 1",
         "runners-up", "outputs", "accumulate (issue)", "final drain"]
NAMES_LDS = ["", "tiles 2..", "merge + certify", "exact evaluation", "outputs", "limb conversion", "atomics (issue)",
             "request next frames", "tile 0", "", "tile 1"]
frames = e.synth.synth_frames(20244, 20, P, 0, S)
out = (C.c_ulonglong * 32)()
with e.VqSession(P) as s:
    s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, 128)
    s.enable_timing(True)
    for M in (256, 512, 1024):
        s.grow()
        dprev = s.prev_distortion()
        for p in range(8):
            lib.e2vq_debug_pre_stamps(None, 1)
            s.run_pass(); s.synchronize()
            ms = s.last_pass_kernel_ms()
            pre, fb = s.last_pass_info()
            lib.e2vq_debug_pre_stamps(out, 0)
            st = s.pass_stats()
            for base, names in ((0, NAMES), (16, NAMES_LDS)):  # k_pass_pre / k_pass_pre_lds
                v = np.array(out[base:base + 16], dtype=np.float64)
                if v[13] == 0:
                    continue
                nblk, nwav = max(v[12], 1), max(v[13], 1)
                tot = v[:12].sum()
                print(f"M={M} pass {p}: kernel {ms:.3f} ms {'k_pass_pre_lds' if base else 'k_pass_pre'} fallback={fb} blocks={int(v[12])} "
                      f"waves={int(v[13])} cycles/block {tot / nblk:.0f} ({tot / nwav / 1e3:.0f} kcyc per wave)")
                print("    " + "  ".join(f"{names[k]}: {v[k] / nblk:.0f}" for k in range(len(names)) if names[k]) +
                      f"  | final drain per wave: {v[9] / nwav:.0f}", flush=True)
            if not pre:
                print(f"M={M} pass {p}: kernel {ms:.3f} ms (plain sweep)", flush=True)
            ratio = (dprev - st.DD) / st.DD
            dprev = st.DD
            if p > 0 and not ratio >= 0.05:
                break
            s.update()
        s.set_prev_distortion(dprev)
