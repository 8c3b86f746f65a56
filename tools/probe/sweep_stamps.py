"""Where a wave of k_sweep_cand spends its cycles (round 5): run against a library built with -DE2VQ_SWEEP_STAMP=1
(tools/probe/ab/build_variant.sh stamp '-DE2VQ_SWEEP_STAMP=1'; ECOZ2VQ_LIB=tools/probe/ab/stamp/libecoz2vq.so).
Prints, per pass of the M = 256 / 512 / 1024 levels on the bench data, the cycles per block and wave by phase."""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import ecoz2rs_amd as e

P, S = 36, int(os.environ.get("STAMP_FRAMES", str(1 << 21)))
os.environ["ECOZ2_VQ_QUIET"] = "1"
fn = e.lib.e2vq_debug_sweep_stamps
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
frames = e.synth.synth_frames(20244, 20, P, 0, S)
with e.VqSession(P) as s:
    s.set_frames(frames)
    s.prepare()
    s.init_codebook()
    s.learn(0.05, 64)
    names = ["wait for B", "stage 1", "stage 2", "rest of the block (fused: reduction)", "rows request", "evaluation", "outputs"]
    for M in (128, 256, 512, 1024):
        s.grow()
        for p in range(3):
            fn(buf, 1)
            s.enable_timing(True)
            s.run_pass()
            s.synchronize()
            ms = s.last_pass_kernel_ms()
            fn(buf, 0)
            st = s.pass_stats()
            s.update()
            n = max(1, buf[8])
            per = [buf[k] / n for k in range(7)]
            tot = sum(per)
            print(f"M {M:5d} pass {p + 1}: pass kernels {ms:.3f} ms; cycles per block and wave: " +
                  ", ".join(f"{nm} {v:8.1f}" for nm, v in zip(names, per)) + f"; limb conversion {buf[12] / n:8.1f}; total {tot + buf[12] / n:8.1f}; rows added to per block {buf[7] / n:.2f}; "
                  f"flagged jobs {buf[10] / max(1, buf[11]):.3f}", flush=True)
        s.set_prev_distortion(st.DD)
