import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ecoz2rs_amd as e
P, M, T = 36, 1024, 10_000_000
frames = e.synth.synth_frames(20243, 20, P, 0, T)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P)
s.set_frames(frames[: 1 << 20]); s.prepare(); s.init_codebook(); s.learn(0.05, M)   # a real M=1024 codebook
t0 = time.perf_counter(); sym, dmin = s.quantize(frames); t1 = time.perf_counter()
print(f"config 3 host-buffer quantize (pageable H2D + re-layout + sweep + D2H): {T/(t1-t0)/1e6:.1f} M frames/s ({(t1-t0)*1e3:.0f} ms)")
d = torch.from_numpy(frames[: 1 << 22]).cuda(); n = d.shape[0]
dsym = torch.empty(n, dtype=torch.int16, device="cuda"); ddm = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); s.quantize_device(d, n, dsym, ddm); s.synchronize(); t1 = time.perf_counter()
    print(f"device-resident quantize of {n} frames (re-layout + sweep): {n/(t1-t0)/1e6:.1f} M frames/s ({(t1-t0)*1e3:.2f} ms)")
