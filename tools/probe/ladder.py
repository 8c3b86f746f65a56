import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
P, S = 36, 1 << 21
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook()
s.enable_timing(True)
flop = lambda M: 2.0 * M * 37 * S
t_all = time.perf_counter()
while s.codebook_size() < 1024:
    s.grow(); M = s.codebook_size()
    ks, t0 = [], time.perf_counter()
    for it in range(3):
        s.run_pass(); ks.append(s.last_pass_kernel_ms()); st = s.pass_stats(); s.update()
    s.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    k = min(ks)
    print(f"M={M:5d} kernel {k:7.3f} ms  step {dt:7.3f} ms  useful {flop(M)/k*1e-9:6.2f} TF  HBM-alg {S*306/k*1e-6:7.1f} GB/s  avg_dist {st.avg_distortion:.4f}")
print("3 passes per level, total wall", round((time.perf_counter() - t_all) * 1e3, 1), "ms")
