"""The prefiltered pass at M = 2048 .. 8192 (the key keeps 22 - log2 M mantissa bits of the value): kernel ms per pass and
frames left to the FP64 fallback sweep, against the plain FP64 sweep; 2^21 frames, P = 36."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
frames = e.synth.synth_frames(20244, 20, P, 0, S)
res = {}
for pre in ("1", "0"):
    os.environ["ECOZ2_VQ_PREFILTER"] = pre
    with e.VqSession(P) as s:
        s.set_frames(frames); s.prepare(); s.init_codebook()
        s.learn(0.05, 1024)
        for M in (2048, 4096, 8192):
            s.enable_timing(True); s.synchronize()
            lv = s.learn(0.05, M)[0]; s.synchronize()
            kms, kn = s.timing_total()
            used, nfb = s.last_pass_info()
            res[(pre, M)] = (lv.passes, kms / max(1, kn), used, nfb, lv.DD)
            print(f"prefilter={pre} M={M}: {lv.passes} passes x {kms / max(1, kn):.3f} ms (sweep + accumulate kernels), "
                  f"prefiltered={used}, fallback frames of the last pass {nfb} ({100.0 * nfb / S:.2f} %), DD={lv.DD!r}", flush=True)
for M in (2048, 4096, 8192):
    a, b = res[("1", M)], res[("0", M)]
    print(f"M={M}: same passes and DD: {a[0] == b[0] and a[4] == b[4]}; speed-up {b[1] / a[1]:.2f}x")
