"""HMM kernels: scoring (sequences x models) and Baum-Welch E+M steps at config-5-like sizes, and -- round 4 -- beyond 64
states (a workgroup per sequence; up to 141 states with the transition matrix staged in LDS, global memory above)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ecoz2rs_amd as e
rng = np.random.default_rng(0)
CASES = ((5, 1024, 20, 2000, 300, 10), (5, 1024, 20, 20000, 300, 10), (32, 1024, 20, 2000, 300, 10), (64, 256, 18, 2000, 300, 10),
         (65, 256, 8, 1024, 300, 3), (96, 256, 8, 1024, 300, 3), (128, 256, 8, 1024, 300, 3), (141, 256, 8, 1024, 300, 3),
         (142, 256, 8, 1024, 300, 3), (192, 256, 4, 512, 300, 3), (256, 256, 4, 512, 300, 3), (512, 256, 2, 256, 300, 2))
for N, M, K, S, T, steps in CASES:
    e.hmm.set_random_seed(1)
    models = [e.hmm.init_model(N, M, 3) for _ in range(K)]
    seqs = [rng.integers(0, M, T).astype(np.uint16) for _ in range(S)]
    e.hmm.score(models[:2], seqs[:10])  # warm-up (module load)
    t0 = time.perf_counter(); r = e.hmm.score(models, seqs); dt = time.perf_counter() - t0
    print(f"score N={N} M={M}: {S} sequences x {K} models x T={T}: {dt*1e3:.1f} ms incl. uploads = "
          f"{S*K*T/dt/1e9:.4f} G symbol-steps/s", flush=True)
    pi, A, B = models[0]
    n = min(S, 2000)
    t0 = time.perf_counter(); _p, _A, _B, hist = e.hmm.train(pi, A, B, seqs[:n], 1e-5, 0.0, steps); dt = time.perf_counter() - t0
    print(f"train N={N} M={M}: {steps} E+M steps over {n} sequences of T={T}: {dt*1e3:.1f} ms = {steps*n*T/dt/1e6:.2f} M symbol-steps/s",
          flush=True)
