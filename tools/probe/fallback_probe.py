#!/usr/bin/env python3
"""Share of the frames the prefiltered pass cannot certify (they go to the FP64 fallback sweep), per level and pass, for several
generators and shard sizes.   usage: fallback_probe.py [T ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ecoz2rs_amd as e  # noqa: E402

os.environ["ECOZ2_VQ_QUIET"] = "1"
P = 36
Ts = [int(x) for x in sys.argv[1:]] or [100000, 1 << 21]
GENS = [("continuum", 1, 6, 0.01), ("continuum, noise 0.002", 1, 6, 0.002), ("20 classes", 0, 20, 0.05), ("20 classes, 4x noise", 0, 20, 0.2)]
for T in Ts:
    for name, kind, ncls, noise in GENS:
        frames = e.synth.synth_frames_kind(20244, kind, ncls, noise, P, 0, T)
        with e.VqSession(P) as s:
            s.set_frames(frames)
            s.prepare()
            s.init_codebook()
            s.learn(0.05, 64)
            m = 128
            while m <= 1024:
                s.grow()
                line = []
                for i in range(3):
                    s.synchronize()
                    t0 = time.perf_counter()
                    s.run_pass()
                    st = s.pass_stats()
                    s.synchronize()
                    dt = time.perf_counter() - t0
                    kd, two, ff = s.last_pass_sweep()
                    pre, nfb = s.last_pass_info()
                    line.append(f"pass {i}: kind {kd}/{int(two)} uncertified {nfb / T:.4f} step {dt * 1e3:.3f} ms avg {st.avg_distortion:.4f}")
                    s.update()
                print(f"{name:24s} T={T} M={m}: " + "; ".join(line), flush=True)
                m *= 2
