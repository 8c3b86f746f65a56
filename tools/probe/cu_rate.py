"""k_cell_update in isolation: a short ladder to M = 1024 on 2^18 frames, then 12 pass / statistics / update rounds.
Run under `rocprofv3 --kernel-trace` and read the kernel's durations (tools/summarize_trace.py), or take the wall
time per round printed here.  ECOZ2VQ_LIB selects an A/B variant (tools/probe/ab)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
os.environ["ECOZ2_VQ_QUIET"] = "1"
P, S = 36, 1 << 18
frames = e.synth.synth_frames(20244, 20, P, 0, S)
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook()
s.learn(0.05, 1024)
for M in (1024, 64):
    if M == 64:
        s.init_codebook(); s.learn(0.05, 64)
    t0 = time.perf_counter()
    for it in range(12):
        s.run_pass(); s.pass_stats(); s.update()
    s.synchronize()
    print(f"M={M}: {(time.perf_counter() - t0) / 12 * 1e6:.1f} us per round ({os.environ.get('ECOZ2VQ_LIB', 'product')})")
