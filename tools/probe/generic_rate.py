"""Sweep rate of the larger prediction orders, M = 1024: P = 41 .. 80 on the FP64 matrix pipe (round 4: half blocks per wave),
beyond that k_pass_generic_lds."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
os.environ["ECOZ2_VQ_QUIET"] = "1"
M, T = 1024, 1 << 18
for P in (40, 48, 64, 80, 100, 200):
    frames = e.synth.synth_frames(7, 8, P, 0, T)
    rng = np.random.default_rng(P)
    refl = np.zeros((M, P + 1)); refl[:, 1:] = rng.uniform(-0.3, 0.3, (M, P)) * 0.9 ** np.arange(P)
    with e.VqSession(P) as s:
        s.set_frames(frames); s.prepare(); s.set_codebook(refl)
        s.enable_timing(True)
        ts = []
        for _ in range(4):
            s.run_pass(); ts.append(s.last_pass_kernel_ms()); s.pass_stats(); s.update()
        ms = min(ts)
        print(f"P={P}: pass kernel {ms:.2f} ms on {T} frames x {M} codewords = {2 * M * (P + 1) * T / ms * 1e-9:.1f} TFLOP/s useful", flush=True)
