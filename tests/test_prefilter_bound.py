"""CPU check of the mathematics behind the prefiltered sweep (ecoz2rs_amd/csrc/vq_prefilter.hip).

A numpy restatement of the limb split, the two exact integer sums and the f32 key, on synthetic frames and
codebooks of several kinds, checks what the kernel relies on:
  * every limb is within its range and every |W| partial-sum bound stays below 2^24 (exact f32 accumulation),
  * |2^27 sum(xi eta) - key| <= 2^8 (sum|xi| + max sum|eta| + NC/2 + 1) + |key| 2^-(22 - idxbits)   (the proven bound),
  * whenever the third key is farther than tau from the first, the true argmin is one of the first two.
No GPU and no product code involved: this pins the host-side logic (scales, tolerance) the kernels implement."""
import numpy as np
import pytest

import ecoz2rs_amd as e

P = 36
NC = P + 1


def _split(x):
    s1 = x * 512.0
    l1 = np.rint(s1)
    s2 = (s1 - l1) * 512.0
    l2 = np.rint(s2)
    return l1, l2


def _ilogb(a):
    return np.frexp(a)[1] - 1


def _keys(frames, cq, ea):
    """Returns (v as float32, g per frame, ymax, scale per frame) following k_pre_frames / k_pre_codebook."""
    nz = frames != 0.0
    e_fr = np.where(nz, _ilogb(np.where(nz, frames, 1.0)) - ea[None, :] + 1, -100000)
    eA = e_fr.max(axis=1)
    eA = np.where(eA == -100000, 0, eA)
    xi = np.ldexp(frames, (-ea[None, :] - eA[:, None]).astype(np.int64))
    nzc = cq != 0.0
    eC = (np.where(nzc, _ilogb(np.where(nzc, cq, 1.0)) + ea[None, :] + 1, -100000)).max()
    eta = np.ldexp(cq, (ea[None, :] - eC).astype(np.int64))
    assert np.abs(xi).max() < 1.0 and np.abs(eta).max() < 1.0
    X, Y = _split(xi), _split(eta)
    for L, lim in zip(X + Y, (512, 256) * 2):
        assert np.abs(L).max() <= lim
    W0 = X[0] @ Y[0].T
    W1 = X[0] @ Y[1].T + X[1] @ Y[0].T
    # bounds on every partial sum (sums of absolute values): exact in f32 whatever the summation order
    assert (np.abs(X[0]) @ np.abs(Y[0]).T).max() < 2 ** 24
    assert (np.abs(X[0]) @ np.abs(Y[1]).T + np.abs(X[1]) @ np.abs(Y[0]).T).max() < 2 ** 24
    v = (W0 * 512.0 + W1).astype(np.float32)  # one rounding: what fmaf does (the exact value is an integer < 2^34)
    g = np.abs(xi).sum(axis=1)
    ymax = np.abs(eta).sum(axis=1).max()
    exact = (xi @ eta.T) * 2.0 ** 27  # 2^27 sum xi eta (float64: error ~1e-16 relative of sum |terms|, negligible)
    return v, exact, g, ymax


def _codebook(oracle, src):
    refl = np.zeros((src.shape[0], NC))
    for i in range(src.shape[0]):
        st, _pe, rc, _a = oracle.lpca_r(src[i], P)
        assert st == 0
        refl[i, 1:] = rc[1:]
    return refl


@pytest.mark.parametrize("kind", ["plain", "rescaled", "twins", "codebook_scales"])
def test_limb_prefilter_bound_and_certification(oracle, kind):
    rng = np.random.default_rng(5)
    T, M = 1500, 256
    frames = e.synth.synth_frames(20260, 6, P, 0, T)
    if kind == "rescaled":
        frames = frames * 10.0 ** rng.integers(-9, 9, size=T)[:, None]
    refl = _codebook(oracle, e.synth.synth_frames(20261, 5, P, 0, M))
    if kind == "twins":
        refl = oracle.grow(refl[: M // 2])
    cq = oracle.reflections_to_cq(refl)
    if kind == "codebook_scales":  # vq quantize: a_n from the codebook
        ea = -(_ilogb(np.abs(cq).max(axis=0)) + 1)
    else:  # vq learn: a_n from the data
        ea = _ilogb(np.abs(frames).max(axis=0)) + 1
    v, exact, g, ymax = _keys(frames, cq, ea.astype(np.int64))
    bits = 8  # log2 M
    mask = np.uint32(~((1 << bits) - 1) & 0xFFFFFFFF)
    key = ((v.view(np.uint32) & mask) | np.arange(M, dtype=np.uint32)[None, :]).view(np.float32).astype(np.float64)
    CONST = 0.5 * NC + 1.0
    eps = 256.0 * (g[:, None] + ymax + CONST) + np.abs(key) * 2.0 ** -(22 - bits)
    assert (np.abs(exact - key) <= eps).all(), "the proven bound does not hold"
    # certification as in k_pass_pre
    order = np.argsort(key, axis=1, kind="stable")
    k = np.take_along_axis(key, order[:, :3], axis=1)
    tau = 1.27 * (512.0 * (g + ymax + CONST) + 2.0 * 2.0 ** -(22 - bits) * k[:, 0])
    cert = (k[:, 0] >= 1e-30) & (k[:, 2] > k[:, 0] + tau)
    d = frames @ cq.T  # the distortions themselves (any tie inside float64 noise would sit inside tau anyway)
    best = d.argmin(axis=1)
    in_top2 = (best == order[:, 0]) | (best == order[:, 1])
    assert in_top2[cert].all(), "a certified frame lost its argmin"
    assert cert.mean() > (0.5 if kind != "twins" else 0.2)  # the prefilter certifies most ordinary frames
