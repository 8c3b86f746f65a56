"""Timing experiments on the stamped fused sorted pass (round 5): what a wave's stage 1 / tail cost without one of their
ingredients.  Needs a -DE2VQ_SWEEP_STAMP=2 library (tools/probe/ab/build_variant.sh); results under an experiment are WRONG
by construction -- the level's state is restored before each run.  EXP_M = codebook size(s) to look at."""
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
fx = e.lib.e2vq_debug_sweep_exp
fx.argtypes = [C.c_int]
buf = (C.c_ulonglong * 16)()
frames = e.synth.synth_frames(20244, 20, P, 0, S)
names = ["wait B", "stage 1", "stage 2", "keys loop", "rows request", "evaluation", "outputs"]
modes = [(int(m), "") for m in os.environ.get("EXP_MODES", "0,2,3,6,10,18,31,0").split(",")]
want = [int(x) for x in os.environ.get("EXP_M", "256,1024").split(",")]
for M in want:
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, M // 2)
        s.grow()
        # the level's own first two passes through e2vq_learn's path are seeded / incremental; by hand: one full pass (not
        # grouped), then incremental ones on the sorted list
        for p in range(2):
            s.run_pass()
            s.synchronize()
            st = s.pass_stats()
            s.update()
            s.set_prev_distortion(st.DD)
        s.save_state()
        for mode, what in modes:
            s.restore_state()
            fx(mode)
            fn(buf, 1)
            s.enable_timing(True)
            s.run_pass()
            s.synchronize()
            ms = s.last_pass_kernel_ms()
            fn(buf, 0)
            fx(0)
            n = max(1, buf[8])
            per = [buf[k] / n for k in range(7)]
            print(f"M {M:5d} exp {mode:2d} {what:40s}: kernel {ms:.3f} ms; " + ", ".join(f"{nm} {v:7.0f}" for nm, v in zip(names, per)) +
                  f", conversion {buf[12] / n:6.0f}; total {sum(per) + buf[12] / n:7.0f}; flagged {buf[10] / max(1, buf[11]):.3f}; kind {s.last_pass_sweep()}", flush=True)
