"""Recorded accumulate vs the burst of atomics as a level converges, back to back inside the library: one level run to a small
epsilon (many passes) through e2vq_learn, kernel ms (sweep + reduce) and wall ms per pass for
ECOZ2_VQ_ACCUMULATE=records without the switch (FEW_DIV=0), with it (FEW_DIV=8 / 16 / 32), and ECOZ2_VQ_ACCUMULATE=burst."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
P, S = 36, 1 << 21
os.environ["ECOZ2_VQ_QUIET"] = "1"
frames = e.synth.synth_frames(20244, 20, P, 0, S)
EPS = float(sys.argv[1]) if len(sys.argv) > 1 else 0.002
for M in (1024, 512, 256):
    DIVS = [int(x) for x in os.environ.get("CROSSOVER_DIVS", "0,8,16,32").split(",")]
    modes = [(f"records, switch below 1/{d}" if d else "records, no switch", {"ECOZ2_VQ_ACCUMULATE": "records", "ECOZ2_VQ_RECORDS_FEW_DIV": str(d)}) for d in DIVS]
    modes.append(("burst", {"ECOZ2_VQ_ACCUMULATE": "burst", "ECOZ2_VQ_PREFILTER_MIN_M": "128"}))
    for mode, env in modes:
        os.environ.pop("ECOZ2_VQ_PREFILTER_MIN_M", None)
        os.environ.update(env)
        with e.VqSession(P) as s:
            s.set_frames(frames); s.prepare(); s.init_codebook(); s.learn(0.05, M // 2)
            best = None
            for rep in range(2):
                s.save_state() if rep == 0 else s.restore_state()
                s.enable_timing(True); s.synchronize()
                t0 = time.perf_counter(); lv = s.learn(EPS, M)[0]; s.synchronize(); wall = time.perf_counter() - t0
                kms, kn = s.timing_total()
                rec, n = s.last_pass_records()
                r = (lv.passes, kms / kn, wall / lv.passes * 1e3, rec, n)
                best = r if best is None or r[2] < best[2] else best
            print(f"M={M} {mode:24s}: {best[0]} passes x kernels {best[1]:.4f} ms, wall {best[2]:.4f} ms per pass; last pass recorded={best[3]}, "
                  f"last record count {best[4]} ({100.0 * best[4] / S:.1f} % of frames)", flush=True)
