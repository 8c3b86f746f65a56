#!/usr/bin/env python3
"""CPU model of the prefilter's keys on data of several shapes: what share of the frames the three-limb
keys can certify, and what each way of gaining precision would be worth -- before any of them is built.

Per generator: a codebook from the strict oracle's ladder on 100 000 frames (M = 1024), keys restated in numpy exactly as the
kernels compute them (tests/test_prefilter_bound.py: limb split, exact integer sums, f32 key, index bits), evaluated on frames
the codebook was NOT trained on (a 2^21-frame training set behaves like those: 2 000 frames per cell).  Prints
  * the cancellation of d = sum r cq at the nearest codeword, tau / k1, the share of (g + y + NC + 4) that is the NC + 4,
  * certified share with the per-coefficient scales a_n from the column max / mean / RMS / balanced against the codebook,
  * certified share when the top 2 / 3 / 4 keys are evaluated exactly, with three weight levels (built) and four,
  * ... with three candidates behind the fourth-key BOUND of the two lane halves (what round 6 built),
  * ... with per-tile codeword scales (lower-bound keys).
Needs no GPU: the oracle library and numpy.   usage: key_precision_model.py [generator ...]   (continuum bench noise4x)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ecoz2rs_amd as e  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.test_prefilter_bound import _ilogb, _split  # noqa: E402

P, NC, M, BITS = 36, 37, 1024, 10
GENS = {"continuum": (1, 6, 0.01), "bench": (0, 20, 0.05), "noise4x": (0, 20, 0.2)}


def keys(frames, cq, ea, levels=3):
    nz = frames != 0.0
    eA = np.where(nz, _ilogb(np.where(nz, frames, 1.0)) - ea[None, :] + 1, -100000).max(axis=1)
    xi = np.ldexp(frames, (-ea[None, :] - eA[:, None]).astype(np.int64))
    eC = (_ilogb(np.where(cq != 0, cq, 1.0)) + ea[None, :] + 1)[cq != 0].max()
    eta = np.ldexp(cq, (ea[None, :] - eC).astype(np.int64))
    X, Y = _split(xi), _split(eta)
    v = (X[0] @ Y[0].T) * 262144.0 + (X[0] @ Y[1].T + X[1] @ Y[0].T) * 512.0 + (X[0] @ Y[2].T + X[1] @ Y[1].T + X[2] @ Y[0].T)
    if levels == 4:
        v = v + (X[1] @ Y[2].T + X[2] @ Y[1].T) / 512.0
    v = v.astype(np.float32)
    mask = np.uint32(~((1 << BITS) - 1) & 0xFFFFFFFF)
    key = ((v.view(np.uint32) & mask) | np.arange(cq.shape[0], dtype=np.uint32)[None, :]).view(np.float32).astype(np.float64)
    return key, xi, eta


def report(name):
    kind, ncls, noise = GENS[name]
    o = oracle_lib.load()
    train = e.synth.synth_frames_kind(20244, kind, ncls, noise, P, 0, 100000)
    rc, levels, _ = o.learn(train, 0.05, M)
    assert rc == 0
    cq = o.reflections_to_cq(levels[-1]["reflections"])
    big = e.synth.synth_frames_kind(20244, kind, ncls, noise, P, 0, 1 << 21)
    x = big[200000::450][:4000]  # frames outside the training set
    d = x @ cq.T
    j = d.argmin(axis=1)
    print(f"== {name}: mean r0 {x[:, 0].mean():.2f}, avg d - 1 at M = {M}: {(d.min(axis=1) - 1).mean():.4f}; "
          f"cancellation sum|r cq| / d at the nearest codeword: median {np.median((np.abs(x) @ np.abs(cq).T)[np.arange(len(x)), j] / d.min(axis=1)):.0f}")
    scales = {
        "a_n = column max (the product)": _ilogb(np.abs(big).max(axis=0)) + 1,
        "a_n = column mean": _ilogb(np.abs(big).mean(axis=0)) + 1,
        "a_n = column RMS": _ilogb(np.sqrt((big ** 2).mean(axis=0))) + 1,
        "a_n balanced sqrt(mean|r| / mean|cq|)": np.round(0.5 * (np.log2(np.abs(big).mean(axis=0)) - np.log2(np.abs(cq).mean(axis=0)))),
    }
    half = ((np.arange(M) >> 2) & 1).astype(bool)
    for label, ea in scales.items():
        ea = ea.astype(np.int64)
        for lv in (3, 4):
            key, xi, eta = keys(x, cq, ea, lv)
            g, y = np.abs(xi).sum(axis=1), np.abs(eta).sum(axis=1).max()
            k = np.sort(key, axis=1)[:, :5]
            const = 41.0 if lv == 3 else 2.0  # (the dropped level-3 products are what the NC + 4 pays for)
            tau = 1.27 * (512.0 * (g + y + const) + 2.0 * 2.0 ** -(22 - BITS) * k[:, 0])
            c = [float((k[:, n] > k[:, 0] + tau).mean()) for n in (2, 3, 4)]
            ka, kb = np.sort(key[:, ~half], axis=1)[:, :3], np.sort(key[:, half], axis=1)[:, :3]
            b4 = np.minimum(np.minimum(ka[:, 2], kb[:, 2]), np.maximum(ka[:, 1], kb[:, 1]))
            c3b = float(((k[:, 2] > k[:, 0] + tau) | (b4 > k[:, 0] + tau)).mean())
            if label.startswith("a_n = column max") or lv == 3:
                print(f"   {label:40s} {lv} levels: tau/k1 {np.median(tau / k[:, 0]):.4f} (NC + 4 is {const / np.median(g + y + const):.2f} of g + y + const); "
                      f"certified with top 2 / 3 / 4 exact: {c[0]:.3f} / {c[1]:.3f} / {c[2]:.3f}; top 3 behind the two-halves bound: {c3b:.3f}")
    # per-tile codeword scales; "lower-bound keys" (every key minus its own tile's tolerance: one more
    # VALU operation per value) are not built; "as built" = only the SMALLEST key takes its tile's
    # tolerance, every other key the codebook-wide one (k_pre_codebook's table, read once per frame)
    ea = scales["a_n = column max (the product)"].astype(np.int64)
    key, xi, eta = keys(x, cq, ea, 3)
    g = np.abs(xi).sum(axis=1)
    e_t = np.repeat(np.ceil(np.log2(np.abs(eta).reshape(M // 32, 32, -1).max(axis=(1, 2)))), 32)  # <= 0: bits a tile leaves unused
    y_t = (np.abs(eta) / 2.0 ** e_t[:, None]).sum(axis=1)
    tol = 1.27 * 256.0 * (g[:, None] + y_t[None, :] + 41.0) * 2.0 ** e_t[None, :]
    lo = key - tol
    best = key.argmin(axis=1)
    hi = (key + tol)[np.arange(len(x)), best]
    n_within = (lo <= hi[:, None]).sum(axis=1)
    print(f"   per-tile codeword scales (tile exponents below the global one: quartiles {np.quantile(-e_t, [0.25, 0.5, 0.75])}), lower-bound keys: "
          f"<= 2 codewords within reach {np.mean(n_within <= 2):.3f}, <= 3: {np.mean(n_within <= 3):.3f}")
    st = np.clip(-e_t, 0, 8)
    y_b = np.abs(eta * 2.0 ** st[:, None]).sum(axis=1).reshape(M // 32, 32).max(axis=1).repeat(32)
    y = np.abs(eta).sum(axis=1).max()
    relk = 2.0 * 2.0 ** -(22 - BITS)
    k = np.sort(key, axis=1)[:, :3]
    t1 = k[:, 0]
    tau_old = 1.27 * (512.0 * (g + y + 41.0) + relk * t1)
    tau_new = 1.27 * (256.0 * 2.0 ** -st[best] * (g + y_b[best] + 41.0) + 256.0 * (g + y + 41.0) + relk * t1)
    half = ((np.arange(M) >> 2) & 1).astype(bool)
    ka, kb = np.sort(key[:, ~half], axis=1)[:, :3], np.sort(key[:, half], axis=1)[:, :3]
    b4 = np.minimum(np.minimum(ka[:, 2], kb[:, 2]), np.maximum(ka[:, 1], kb[:, 1]))
    cert = lambda tau: float(((k[:, 2] > t1 + tau) | (b4 > t1 + tau)).mean())
    print(f"   as built (the smallest key with its tile's tolerance, the others with the codebook's; top 3 behind the two-halves bound): "
          f"certified {cert(tau_old):.3f} -> {cert(tau_new):.3f}, tau {np.median(tau_new / tau_old):.2f} of the codebook-wide one")


for name in (sys.argv[1:] or list(GENS)):
    report(name)
