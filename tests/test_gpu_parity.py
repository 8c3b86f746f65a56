"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle -- bit-exact."""
import numpy as np
import pytest

import ecoz2rs_amd as e
from tests import oracle_lib

pytestmark = pytest.mark.gpu

P = 36


def _frames(seed, T, classes=8):
    return e.synth.synth_frames(seed, classes, P, 0, T)


def _codebook(oracle, frames, M, seed=0):
    """A plausible codebook: reflections of M random frames (lpca_r on their autocorrelation)."""
    rng = np.random.default_rng(seed)
    idx = rng.choice(frames.shape[0], size=M, replace=False)
    refl = np.zeros((M, P + 1))
    for i, t in enumerate(idx):
        st, _pe, rc, _a = oracle.lpca_r(frames[t], P)
        assert st == 0
        refl[i, 1:] = rc[1:]
    return refl


@pytest.mark.parametrize("T,M", [(1000, 2), (4096, 16), (5000, 64), (10000, 128), (7777, 256), (20000, 1024)])
def test_quantize_bit_exact(oracle, T, M):
    frames = _frames(20243, T)
    refl = _codebook(oracle, frames, M)
    cq = oracle.reflections_to_cq(refl)
    sym_o, dmin_o = oracle.quantize(cq, frames)
    with e.VqSession(P) as s:
        s.set_codebook(refl)
        sym, dmin = s.quantize(frames)
    assert np.array_equal(sym, sym_o)
    assert np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))


@pytest.mark.parametrize("T,M", [(1000, 2), (4096, 16), (10000, 128), (7777, 256), (20000, 1024), (9000, 2048), (6000, 4096)])
def test_pass_rows_bit_exact(oracle, T, M):
    """One LBG pass: per-cell exact sums, counts and distortion sums equal the oracle's."""
    frames = _frames(20242, T)
    refl = _codebook(oracle, frames, M, seed=1)
    cq = oracle.reflections_to_cq(refl)
    rc, st = oracle.data_stats(frames)
    assert rc == 0
    sh_r, sh_q = oracle.shifts(st.maxabs)
    Ed = oracle.dist_exponent(cq, st.maxabs)
    _sym_o, _dmin_o, rows_o = oracle.run_pass(cq, frames, sh_r, Ed)
    Q = oracle.unfix(st.q_hi, st.q_lo, sh_q)
    ls_o = oracle.rows_stats(rows_o, P, T, sh_r, Ed, Q)
    refl_o, _failed = oracle.update(rows_o, P, sh_r, refl)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.run_pass()
        rows = s.get_rows()
        ls = s.pass_stats()
        s.update()
        refl_g = s.get_codebook()
    assert oracle_lib.rows_match(rows, rows_o, P)
    assert ls.DD == ls_o.DD and ls.avg_distortion == ls_o.avg and ls.sigma == ls_o.sigma
    assert ls.inertia == ls_o.inertia and ls.empty_cells == ls_o.empty_cells
    assert np.array_equal(refl_g.view(np.uint64), refl_o.view(np.uint64))


@pytest.mark.parametrize("T,maxM", [(10000, 16), (30000, 256)])
def test_learn_ladder_bit_exact(oracle, T, maxM, tmp_path):
    """Whole LBG ladder: every level's codebook, pass count and callback scalars equal the oracle's."""
    frames = _frames(20241, T, classes=4)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, maxM)
    assert rc == 0
    cbs = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        got = []

        def cb(M, avg, sigma, inertia):
            cbs.append((M, avg, sigma, inertia))
            got.append(s.get_codebook())

        levels = s.learn(0.05, maxM, out_root=str(tmp_path), callback=cb)
    assert [l.M for l in levels] == [l["M"] for l in levels_o]
    assert [l.passes for l in levels] == [l["passes"] for l in levels_o]
    for g, o in zip(got, levels_o):
        assert np.array_equal(g.view(np.uint64), o["reflections"].view(np.uint64))
    assert cbs == cbs_o
    # files written by the product read back identical to the oracle's codebooks
    for o in levels_o:
        _cls, _P, refl = e.formats.read_cbook(str(tmp_path / "data" / "codebooks" / "_" / f"eps_0.05_M_{o['M']:04d}.cbook"))
        assert np.array_equal(refl.view(np.uint64), o["reflections"].view(np.uint64))


@pytest.mark.parametrize("Pn", [36, 70])
def test_failed_cells_are_counted_like_the_oracle(oracle, Pn):
    """Cells whose Levinson recursion fails (status 2 of src/lpc/lpca_r_rs.rs:37-39) keep their codeword and are counted
    in e2vq_level_stats.failed_cells -- fused wave-per-cell kernel (P = 36) and thread-per-cell kernels (P = 70)."""
    rng = np.random.default_rng(5)
    good = e.synth.synth_frames(9, 2, Pn, 0, 300)
    # r = [1, .9, -.9, 0...]: pe turns negative at k = 2, for the frame and for any sum of such frames
    bad = np.zeros((200, Pn + 1))
    bad[:, 0], bad[:, 1], bad[:, 2] = 1.0, 0.9, -0.9
    bad[:, :3] *= (1.0 + 1e-3 * rng.standard_normal((200, 1)))
    bad *= 40.0  # far from every codeword of the good frames: they gather in cells of their own
    frames = np.concatenate([good, bad])
    M = 4
    refl = np.zeros((M, Pn + 1))
    for i in range(M):
        refl[i, 1:] = oracle.lpca_r(good[i * 50], Pn)[2][1:]
    cq = oracle.reflections_to_cq(refl)
    rc, st = oracle.data_stats(frames)
    assert rc == 0
    sh_r, _ = oracle.shifts(st.maxabs)
    _s, _d, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
    refl_o, failed_o = oracle.update(rows_o, Pn, sh_r, refl)
    assert failed_o >= 1
    with e.VqSession(Pn) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.run_pass()
        assert oracle_lib.rows_match(s.get_rows(), rows_o, Pn)
        ls = s.pass_stats()
        s.update()
        refl_g = s.get_codebook()
    assert ls.failed_cells == failed_o
    assert np.array_equal(refl_g.view(np.uint64), refl_o.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("M,T", [(8192, 30000), (65536, 20000)])
def test_large_codebooks_statistics_and_update(oracle, M, T):
    """Codebooks far beyond the ladder's usual sizes (the API takes M <= 65536): k_cell_update then runs more workgroups
    than the chip holds at once (M / 4 + 1, the last one publishing the statistics once every cell has raised its flag),
    most cells are empty and keep their codeword.  Rows, statistics and the updated codebook against the oracle."""
    frames = e.synth.synth_frames(777, 20, P, 0, T)
    rng = np.random.default_rng(M)
    rc, st = oracle.data_stats(frames)
    sh_r, sh_q = oracle.shifts(st.maxabs)
    rc, levels, _cb = oracle.learn(frames[:6000], 0.05, 64)
    refl = np.tile(levels[-1]["reflections"], (M // 64, 1)) * (1.0 + 1e-3 * rng.standard_normal((M, 1)))
    refl[:, 0] = 0.0
    refl = np.clip(refl, -0.999, 0.999)
    cq = oracle.reflections_to_cq(refl)
    Ed = oracle.dist_exponent(cq, st.maxabs)
    _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, Ed)
    ls_o = oracle.rows_stats(rows_o, P, T, sh_r, Ed, oracle.unfix(st.q_hi, st.q_lo, sh_q))
    refl_o, failed_o = oracle.update(rows_o, P, sh_r, refl)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.run_pass()
        rows = s.get_rows()
        ls = s.pass_stats()
        s.update()
        cb = s.get_codebook()
    assert oracle_lib.rows_match(rows, rows_o, P)
    assert (ls.DD, ls.sigma, ls.inertia, ls.empty_cells, ls.failed_cells) == (ls_o.DD, ls_o.sigma, ls_o.inertia, ls_o.empty_cells, failed_o)
    assert ls.empty_cells > M // 2
    assert np.array_equal(cb.view(np.uint64), refl_o.view(np.uint64))


def test_publish_verification_mode(oracle, monkeypatch):
    """ECOZ2_VQ_VERIFY_PUBLISH=1: after every pass the statistics the update kernel published through host-mapped memory
    while it was still running -- level sums, within-cell terms, L1 maximum, failed recursions -- are recomputed on the host
    from a copy of the rows and compared bit for bit (a lost or early publication would otherwise only show as a different
    convergence decision).  A whole ladder to M = 256 with it on: every pass checked, same codebook as the oracle."""
    import ctypes as C

    monkeypatch.setenv("ECOZ2_VQ_VERIFY_PUBLISH", "1")
    monkeypatch.setenv("ECOZ2_VQ_QUIET", "1")
    frames = e.synth.synth_frames(4242, 6, P, 0, 30011)
    rc, levels_o, _cbs = oracle.learn(frames, 0.05, 256)
    assert rc == 0
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, 256)
        v = C.c_int64()
        e.check(e.lib.e2vq_verified_passes(s._h, C.byref(v)))
        assert v.value == sum(lv.passes for lv in levels) > 20
        assert np.array_equal(s.get_codebook().view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
