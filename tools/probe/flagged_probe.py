#!/usr/bin/env python3
"""What share of the two-stage sweep's (tile, column block) jobs is flagged, and which sweep the host runs afterwards --
for small shards of several generators and for a codebook whose codewords were permuted (tests/test_gpu_prefilter.py's
directed test of the one-stage switch is built on what this prints).   usage: flagged_probe.py [T] [M]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ecoz2rs_amd as e  # noqa: E402

os.environ["ECOZ2_VQ_QUIET"] = "1"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
MAXM = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P = 36
GENS = [("20 classes", 0, 20, 0.05), ("1 class", 0, 1, 0.05), ("200 classes", 0, 200, 0.05), ("20 classes, 4x noise", 0, 20, 0.2),
        ("1 class, noise 0.3", 0, 1, 0.3), ("continuum", 1, 6, 0.01)]
for name, kind, ncls, noise in GENS:
    frames = e.synth.synth_frames_kind(777, kind, ncls, noise, P, 0, T)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, 128)
        m = 256
        while m <= MAXM:
            s.grow()
            line = []
            for i in range(4):
                s.run_pass()
                st = s.pass_stats()
                kd, two, ff = s.last_pass_sweep()
                line.append(f"pass {i}: kind {kd} two {int(two)} flagged {ff:.3f}")
                s.update()
            print(f"{name:24s} T={T} M={m}: " + "; ".join(line), flush=True)
            m *= 2
        # the same codebook with its codewords permuted: neighbours in the index are no longer neighbours in space
        cb = s.get_codebook()
        rng = np.random.default_rng(5)
        perm = rng.permutation(cb.shape[0])
        s.set_codebook(cb[perm])
        line = []
        for i in range(4):
            s.run_pass()
            st = s.pass_stats()
            kd, two, ff = s.last_pass_sweep()
            line.append(f"pass {i}: kind {kd} two {int(two)} flagged {ff:.3f}")
            s.update()
        print(f"{name:24s} T={T} M={cb.shape[0]} PERMUTED: " + "; ".join(line), flush=True)
