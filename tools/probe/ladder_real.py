"""The real ladder 2 .. 1024 on 2^21 frames through e2vq_learn, level by level (a synchronisation per level), with the
event-timed sweep(+accumulate) kernel of every pass: per level passes, kernel ms per pass, wall ms per pass."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
frames = e.synth.synth_frames(20244, 20, P, 0, S)
with e.VqSession(P) as s:
    s.set_frames(frames); s.prepare()
    for rep in range(2):
        s.init_codebook(); s.synchronize()
        out, tot, m = [], 0.0, 2
        while m <= 1024:
            s.enable_timing(True); s.synchronize()
            t0 = time.perf_counter(); lv = s.learn(0.05, m)[0]; s.synchronize(); wall = time.perf_counter() - t0
            kms, kn = s.timing_total()
            out.append(f"M={m}: {lv.passes} x {kms / max(1, kn):.3f} ({wall / lv.passes * 1e3:.3f})")
            tot += wall; m *= 2
        s.enable_timing(False); s.init_codebook(); s.synchronize()
        t0 = time.perf_counter(); s.learn(0.05, 1024); s.synchronize(); one = time.perf_counter() - t0
    print("  ".join(out[:5])); print("  ".join(out[5:]))
    print(f"ladder level by level {tot * 1e3:.2f} ms, in one call {one * 1e3:.2f} ms")
