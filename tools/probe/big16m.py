import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
from tests import oracle_lib
P, M, T = 36, 1024, 1 << 24
t0 = time.time(); frames = e.synth.synth_frames(20244, 20, P, 0, T); print(f"synth {T} frames: {time.time()-t0:.1f}s", flush=True)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P)
t0 = time.time(); s.set_frames(frames); s.prepare(); s.synchronize(); print(f"upload+layout+stats: {time.time()-t0:.2f}s", flush=True)
s.init_codebook()
t0 = time.time(); levels = s.learn(0.05, M); s.synchronize(); dt = time.time() - t0
print(f"learn 2..{M}: {dt*1e3:.0f} ms, passes {[l.passes for l in levels]}, {T/dt/1e6:.1f} M frames/s end to end, final avg distortion {levels[-1].avg_distortion:.6f}", flush=True)
s.enable_timing(True); s.run_pass(); k = s.last_pass_kernel_ms(); rows = s.get_rows()
print(f"M={M} pass kernel on {T} frames: {k:.2f} ms = {T/k*1e-6:.3f} G frames/s, {2*M*37*T/k*1e-9:.1f} TF useful", flush=True)
assert int(rows[:, 74].sum()) == T, "every frame must be counted once"
# sampled oracle check of the assignment against the final codebook
refl = s.get_codebook(); oracle = oracle_lib.load(); cq = oracle.reflections_to_cq(refl)
sl = slice(7_000_000, 7_400_000)
sym, dmin = s.quantize(frames[sl]); so, do = oracle.quantize(cq, frames[sl])
assert np.array_equal(sym, so) and np.array_equal(dmin.view(np.uint64), do.view(np.uint64))
print("count invariant and sampled oracle check ok")
