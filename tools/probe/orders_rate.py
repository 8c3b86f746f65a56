"""Prefiltered vs plain FP64 sweep for several prediction orders: M = 1024 pass over 2^20 resident frames.
(ECOZ2_VQ_PREFILTER is read at session creation, so each measurement opens its own session.)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
os.environ["ECOZ2_VQ_QUIET"] = "1"
T, M = 1 << 20, 1024
for P in (12, 16, 20, 24, 28, 32, 36, 40):
    frames = e.synth.synth_frames(1, 20, P, 0, T)
    res = {}
    for pre in ("1", "0"):
        os.environ["ECOZ2_VQ_PREFILTER"] = pre
        s = e.VqSession(P); s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(1e9, M); s.enable_timing(True)
        ks = []
        for i in range(4):
            s.run_pass(); ks.append(s.last_pass_kernel_ms()); s.pass_stats(); s.update()
        res[pre] = (min(ks[1:]), s.last_pass_info())
        s.close()
    (kp, (used, fb)), (k0, _) = res["1"], res["0"]
    print(f"P={P:2d}: prefiltered {kp:.3f} ms ({T/kp*1e-6:.3f} G frames/s, fallback {fb}, used={used})  plain FP64 {k0:.3f} ms "
          f"({2*M*(P+1)*T/k0*1e-9:.1f} TF)  speed-up {k0/kp:.2f}x", flush=True)
