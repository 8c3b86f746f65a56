"""kernel time of the M <= 32 passes (k_pass_small / LDS-table kernel) on 2^21 frames; A/B builds via ECOZ2VQ_LIB"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ecoz2rs_amd as e
P, S = 36, 1 << 21
frames = e.synth.synth_frames(20244, 20, P, 0, S)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.enable_timing(True)
out = []
while s.codebook_size() < 32:
    s.grow(); M = s.codebook_size()
    ks = []
    for it in range(5):
        s.run_pass(); ks.append(s.last_pass_kernel_ms())
    s.pass_stats(); s.update()
    out.append(f"M={M}: {min(ks):.3f}")
print(os.environ.get("ECOZ2VQ_LIB", "product"), " ".join(out))
