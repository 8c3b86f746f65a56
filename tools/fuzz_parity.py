#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity fuzz (bit-exact): random T, M, P, codebooks; one pass + update + quantize each.
usage: tools/fuzz_parity.py [n_cases] [seed]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ecoz2rs_amd as e
from tests import oracle_lib

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle = oracle_lib.load()
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    bad = 0
    t0 = time.time()
    for case in range(n):
        P = int(rng.choice([36, 36, 36, int(rng.integers(4, 41)), int(rng.integers(4, 41)), int(rng.integers(41, 60))]))
        T = int(rng.choice([rng.integers(1, 200), rng.integers(200, 6000), rng.integers(6000, 40000)]))
        M = int(rng.choice([rng.integers(1, 20), rng.integers(20, 200), rng.integers(200, 700), rng.integers(700, 1500)]))
        frames = e.synth.synth_frames(int(rng.integers(1, 1 << 30)), int(rng.integers(1, 8)), P, int(rng.integers(0, 1000)), T)
        src = e.synth.synth_frames(int(rng.integers(1, 1 << 30)), 5, P, 0, M)
        refl = np.zeros((M, P + 1))
        for i in range(M):
            refl[i, 1:] = oracle.lpca_r(src[i], P)[2][1:] * rng.uniform(0.9, 1.0)
        cq = oracle.reflections_to_cq(refl)
        rc, st = oracle.data_stats(frames)
        sh_r, sh_q = oracle.shifts(st.maxabs)
        Ed = oracle.dist_exponent(cq, st.maxabs)
        sym_o, dmin_o, rows_o = oracle.run_pass(cq, frames, sh_r, Ed)
        ls_o = oracle.rows_stats(rows_o, P, T, sh_r, Ed, oracle.unfix(st.q_hi, st.q_lo, sh_q))
        refl_o, _ = oracle.update(rows_o, P, sh_r, refl)
        with e.VqSession(P) as s:
            s.set_frames(frames); s.prepare(); s.set_codebook(refl)
            sym, dmin = s.quantize(frames)  # against the codebook as given (before the update below)
            s.run_pass(); rows = s.get_rows(); ls = s.pass_stats(); s.update(); refl_g = s.get_codebook()
        ok = (np.array_equal(rows, rows_o) and np.array_equal(sym, sym_o)
              and np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))
              and np.array_equal(refl_g.view(np.uint64), refl_o.view(np.uint64))
              and ls.DD == ls_o.DD and ls.sigma == ls_o.sigma and ls.inertia == ls_o.inertia
              and ls.empty_cells == ls_o.empty_cells)
        if not ok:
            bad += 1
            what = [k for k, v in dict(rows=np.array_equal(rows, rows_o), sym=np.array_equal(sym, sym_o),
                                       dmin=np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64)),
                                       refl=np.array_equal(refl_g.view(np.uint64), refl_o.view(np.uint64)),
                                       DD=ls.DD == ls_o.DD, sigma=ls.sigma == ls_o.sigma,
                                       inertia=ls.inertia == ls_o.inertia, empty=ls.empty_cells == ls_o.empty_cells).items() if not v]
            print(f"MISMATCH case {case}: P={P} T={T} M={M}: {what}", flush=True)
        if case % 25 == 24:
            print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
    print(f"fuzz done: {n} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
