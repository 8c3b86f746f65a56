"""GPU parity of the prefiltered sweep (f16 limb prefilter + exact FP64 evaluation + fallback list).

Every case runs one LBG pass through the C-ABI with the prefilter forced on (from M = 64) and compares symbols,
minimum distortions and the per-cell integer rows bit for bit with the CPU oracle; the adversarial cases are built
to defeat a prefilter that decides anything by itself (ties, twins, zero / negative / badly scaled data)."""
import ctypes as C

import numpy as np
import pytest

import ecoz2rs_amd as e
from tests import oracle_lib

pytestmark = pytest.mark.gpu

P = 36


def _frames(seed, T, classes=8):
    return e.synth.synth_frames(seed, classes, P, 0, T)


def _codebook(oracle, frames, M, seed=0):
    rng = np.random.default_rng(seed)
    idx = rng.choice(frames.shape[0], size=M, replace=frames.shape[0] < M)
    refl = np.zeros((M, P + 1))
    for i, t in enumerate(idx):
        st, _pe, rc, _a = oracle.lpca_r(frames[t], P)
        assert st == 0
        refl[i, 1:] = rc[1:]
    return refl


def _oracle_pass(oracle, frames, refl):
    cq = oracle.reflections_to_cq(refl)
    rc, st = oracle.data_stats(frames)
    assert rc == 0
    sh_r, _ = oracle.shifts(st.maxabs)
    Ed = oracle.dist_exponent(cq, st.maxabs)
    return oracle.run_pass(cq, frames, sh_r, Ed)


class _DeviceBuffer:
    """Device memory through the HIP runtime libecoz2vq.so itself is linked to: dlsym on the library's handle
    searches its dependencies, so no second copy of the runtime (torch bundles one) can get involved."""

    def __init__(self, nbytes):
        self.hip = e.lib
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        assert self.hip.hipSetDevice(0) == 0  # (this thread may never have touched the runtime)
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0
        assert self.hip.hipMemset(self.ptr, 0xFF, C.c_size_t(nbytes)) == 0

    def to_host(self, dtype):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(self.nbytes), 2) == 0  # D2H
        return out

    def from_host(self, array, offset=0):
        a = np.ascontiguousarray(array)
        dst = C.c_void_p(self.ptr.value + offset)
        assert self.hip.hipMemcpy(dst, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0  # H2D

    def free(self):
        self.hip.hipFree(self.ptr)


def _gpu_pass(frames, refl, monkeypatch, prefilter=True):
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER", "1" if prefilter else "0")
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    T = frames.shape[0]
    sym, dmin = _DeviceBuffer(2 * T), _DeviceBuffer(8 * T)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.run_pass(sym.ptr.value, dmin.ptr.value)
        rows = s.get_rows()
        used, nfb = s.last_pass_info()
        s.synchronize()
    assert used == prefilter
    _gpu_pass.fallback = nfb
    out = sym.to_host(np.uint16), dmin.to_host(np.float64), rows
    sym.free()
    dmin.free()
    return out


def _check(oracle, frames, refl, monkeypatch):
    sym_o, dmin_o, rows_o = _oracle_pass(oracle, frames, refl)
    sym, dmin, rows = _gpu_pass(frames, refl, monkeypatch)
    assert np.array_equal(sym, sym_o)
    assert np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))
    assert oracle_lib.rows_match(rows, rows_o, P)


@pytest.mark.parametrize("T", [1, 31, 33, 64, 65, 1000, 8191, 70001])
@pytest.mark.parametrize("M", [2, 4, 8, 16, 32, 64, 128])
def test_small_codebook_register_sums(oracle, monkeypatch, T, M):
    """k_pass_small (M <= 16: cell sums as per-wave register accumulators fed by the i8 matrix unit) and, for M = 32, 64, 128,
    the LDS-table kernel: ragged frame counts (half blocks, one frame, more blocks than waves), with the accumulators flushed
    after every block and only at the end; rows, symbols and distortions equal the oracle's."""
    pool = _frames(20250 + M, max(T, 256))
    refl = _codebook(oracle, pool, M, seed=3)
    frames = np.ascontiguousarray(pool[:T])
    sym_o, dmin_o, rows_o = _oracle_pass(oracle, frames, refl)
    got = []
    for flush_mask in ("65535", "0"):
        monkeypatch.setenv("ECOZ2_VQ_SMALL_FLUSH_MASK", flush_mask)
        sym, dmin = _DeviceBuffer(2 * T), _DeviceBuffer(8 * T)
        with e.VqSession(P) as s:
            s.set_frames(frames)
            s.prepare()
            s.set_codebook(refl)
            s.run_pass(sym.ptr.value, dmin.ptr.value)
            rows = s.get_rows()
            s.synchronize()
        assert oracle_lib.rows_match(rows, rows_o, P), flush_mask
        assert np.array_equal(sym.to_host(np.uint16), sym_o)
        assert np.array_equal(dmin.to_host(np.float64).view(np.uint64), dmin_o.view(np.uint64))
        sym.free()
        dmin.free()
        got.append(rows)
    assert np.array_equal(got[0], got[1])


@pytest.mark.parametrize("T,M", [(5000, 64), (7777, 256), (20000, 1024), (9001, 2048), (6000, 4096), (9000, 8192)])
def test_prefiltered_pass_bit_exact(oracle, monkeypatch, T, M):
    frames = _frames(20250, T)
    _check(oracle, frames, _codebook(oracle, frames, M, seed=3), monkeypatch)
    print(f"T={T} M={M}: {_gpu_pass.fallback} frames to the FP64 fallback")
    assert _gpu_pass.fallback < 0.1 * T  # the prefilter certifies nearly every frame of ordinary data


def test_prefilter_off_gives_the_same_rows(oracle, monkeypatch):
    frames = _frames(20251, 6000)
    refl = _codebook(oracle, frames, 512, seed=4)
    a = _gpu_pass(frames, refl, monkeypatch, prefilter=True)
    b = _gpu_pass(frames, refl, monkeypatch, prefilter=False)
    for x, y in zip(a[:2], b[:2]):  # symbols, distortions
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    assert oracle_lib.rows_match(a[2], b[2], P)  # (the distortion columns: totals only)


def test_twin_and_duplicate_codewords(oracle, monkeypatch):
    """Every codeword appears twice (exact ties: the lower index must win) next to 1 % twins of the LBG split."""
    frames = _frames(20252, 4000)
    base = _codebook(oracle, frames, 64, seed=5)
    refl = np.concatenate([base, base, oracle.grow(base)], axis=0)  # 256 codewords, pairs (i, i + 64) identical
    sym_o, _d, _r = _oracle_pass(oracle, frames, refl)
    assert (sym_o < 64).sum() > 0 and not ((sym_o >= 64) & (sym_o < 128)).any()
    _check(oracle, frames, refl, monkeypatch)
    assert _gpu_pass.fallback > 0  # exact ties between three codewords cannot be certified from two candidates


def test_codebook_made_of_the_frames(oracle, monkeypatch):
    """Many near-ties: the codebook is the LPC of the first 256 frames, and the frames repeat."""
    frames = _frames(20253, 256)
    frames = np.concatenate([frames] * 9 + [frames[:77]], axis=0)
    _check(oracle, frames, _codebook(oracle, frames[:256], 256, seed=6), monkeypatch)


def test_badly_scaled_zero_and_negative_frames(oracle, monkeypatch):
    """Frames spanning 40 orders of magnitude, all-zero frames, sign-flipped frames (negative distortions) and
    coefficients that are zero in every frame: the prefilter must hand what it cannot certify to the FP64 sweep."""
    frames = _frames(20254, 3000)
    rng = np.random.default_rng(7)
    refl = _codebook(oracle, frames, 128, seed=7)
    frames = frames.copy()
    frames[::7] *= 10.0 ** rng.integers(-20, 20, size=frames[::7].shape[0])[:, None]
    frames[5::11] = 0.0
    frames[3::13] *= -1.0
    frames[:, 30] = 0.0
    frames[100:200, 1:] = rng.standard_normal((100, P)) * frames[100:200, :1]
    _check(oracle, frames, refl, monkeypatch)


@pytest.mark.parametrize("wild,bits", [((0.5, 0.8), 12), ((0.3, 0.6), 4)])
@pytest.mark.parametrize("accumulate", ["sorted", "records", "quantize"])
def test_codeword_tiles_of_very_different_scales(oracle, monkeypatch, accumulate, wild, bits):
    """Round 6: the limbs of a 32-codeword tile are split from the codewords scaled by the tile's own power of two (clamped
    to 2^8) and the smallest key is certified with its tile's tolerance.  A codebook whose tiles sit far apart: one tile of
    random large reflections (autocorrelations up to 2^19 resp. 2^10: it sets the codebook-wide scale, 14 resp. 5 bits
    above the ordinary tiles -- beyond resp. within the clamp), one of all-zero reflections (32 identical codewords
    [1, 0, ...]: the clamp, and exact ties), one of damped codewords, the rest ordinary -- through the fused sorted pass,
    round 4's kernel and the fused quantize kernel; the oracle's bits every time."""
    frames = _frames(20256, 6000)
    refl = _codebook(oracle, frames, 256, seed=11)
    rng = np.random.default_rng(11)
    refl[0:32, 1:] = rng.choice([-1.0, 1.0], size=(32, P)) * rng.uniform(wild[0], wild[1], size=(32, P))
    refl[32:64, 1:] = 0.0
    refl[160:192, 1:] *= 0.03
    cq = oracle.reflections_to_cq(refl)
    spread = np.log2(np.abs(cq).reshape(8, -1).max(axis=1))
    assert spread[0] - np.median(spread) > bits
    if accumulate == "quantize":
        T = frames.shape[0]
        buf = _DeviceBuffer(frames.nbytes)
        buf.from_host(frames)
        sym, dmin = _DeviceBuffer(2 * T), _DeviceBuffer(8 * T)
        with e.VqSession(P) as s:
            s.set_frames(frames)
            s.prepare()
            s.set_codebook(refl)
            s.quantize_device(buf.ptr.value, T, sym.ptr.value, dmin.ptr.value)
            s.synchronize()
        sym_o, dmin_o = oracle.quantize(cq, frames)
        assert np.array_equal(sym.to_host(np.uint16), sym_o)
        assert np.array_equal(dmin.to_host(np.float64).view(np.uint64), dmin_o.view(np.uint64))
        for b in (buf, sym, dmin):
            b.free()
        return
    # three passes with updates in between (the fused sorted kernel serves from the second pass of a codebook size on; cells
    # nobody falls into keep their codewords, so the wild tile stays)
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", accumulate)
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER", "1")
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    shares, kinds = [], []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.set_sweep_policy(-1.0, 1.0)  # (the prefiltered kernels whatever they leave uncertified)
        for it in range(3):
            cq = oracle.reflections_to_cq(refl)
            Ed = oracle.dist_exponent(cq, st.maxabs)
            _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, Ed)
            refl, _ = oracle.update(rows_o, P, sh_r, refl)
            s.run_pass()
            used, nfb = s.last_pass_info()
            assert used
            shares.append(nfb / frames.shape[0])
            kinds.append(s.last_pass_sweep()[0])
            assert oracle_lib.rows_match(s.get_rows(), rows_o, P), it
            s.pass_stats()
            s.update()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64)), it
    print(f"{accumulate}: kernel kinds {kinds}, uncertified shares {[round(x, 4) for x in shares]}")
    if accumulate == "sorted":
        assert kinds[1] == 3 and kinds[2] == 3  # the fused pass over frames grouped by cell


def test_identical_frames_and_constant_codebook(oracle, monkeypatch):
    """One frame repeated against a codebook whose codewords differ in the last reflection coefficient only."""
    one = _frames(20255, 1)
    frames = np.repeat(one, 1000, axis=0)
    st, _pe, rc, _a = oracle.lpca_r(one[0], P)
    assert st == 0
    refl = np.zeros((64, P + 1))
    refl[:, 1:] = rc[1:]
    refl[:, P] += np.linspace(-1e-9, 1e-9, 64)
    _check(oracle, frames, refl, monkeypatch)


def test_learn_ladder_with_prefilter_matches_oracle(oracle, monkeypatch):
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER", "1")
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    frames = _frames(20256, 20000, classes=6)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, 512)
    assert rc == 0
    cbs = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, 512, callback=lambda M, a, sg, i: cbs.append((M, a, sg, i)))
        refl = s.get_codebook()
    assert [(l.M, l.passes) for l in levels] == [(l["M"], l["passes"]) for l in levels_o]
    assert cbs == cbs_o
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))


@pytest.mark.parametrize("M,prefilter", [(16, True), (64, False), (256, False), (256, True), (1024, False)])
@pytest.mark.parametrize("accumulate", ["sorted", "sweep", "records", "burst"])
@pytest.mark.parametrize("collective", [False, True])
def test_incremental_rows_match_a_full_accumulation_every_pass(oracle, monkeypatch, collective, accumulate, M, prefilter):
    """Five passes at one codebook size with centroid updates in between: from the second pass on only the frames
    that changed cell are moved (LDS-table, hybrid and global-atomic accumulates of the plain sweep, the prefiltered
    sweep and its fallback list alike); the rows must equal the oracle's full accumulation every time.  `collective`
    routes the rows through the all-reduce hook (own copy + reduced copy)."""
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER", "1" if prefilter else "0")
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    # the accumulates of a prefiltered pass -- "sorted" (round 5, the default from M = 256 on; here from 64): a full first
    # pass as candidate sweep + finishing kernel + k_reduce_records, then the fused pass over frames grouped by cell;
    # "sweep": candidate sweep + finishing kernel + k_reduce_records throughout; "records": round 4's fused kernel, whose
    # sweep records every contribution for k_reduce_records; "burst": round 3's one kernel with its atomics
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", accumulate)
    if collective:
        monkeypatch.setenv("ECOZ2_VQ_FORCE_ALLREDUCE", "1")
    frames = _frames(20257, 9000)
    refl = np.concatenate([_codebook(oracle, frames, M // 2, seed=8)] * 2, axis=0)  # duplicates: a busy fallback list
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    calls = []
    with e.VqSession(P) as s:
        if collective:
            s.set_allreduce(lambda buf, count, op, stream: calls.append(count) or 0, 0, 1)
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        for it in range(5):
            cq = oracle.reflections_to_cq(refl)
            _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
            s.run_pass()
            expect_pre = prefilter and M >= 64
            assert s.last_pass_info()[0] == expect_pre
            assert oracle_lib.rows_match(s.get_rows(), rows_o, P), f"pass {it}"
            refl, _failed = oracle.update(rows_o, P, sh_r, refl)
            s.update()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
    assert bool(calls) == collective


@pytest.mark.parametrize("accumulate", ["sorted", "auto"])
@pytest.mark.parametrize("M,T", [(2048, 9001), (8192, 12000)])
def test_sorted_pass_big_codebooks(oracle, monkeypatch, accumulate, M, T):
    """The fused pass over frames grouped by cell at the large end: 64 and 256 codeword tiles (the list of flagged tiles of a
    turn holds one 16-bit entry per tile: 256 is its capacity, tile numbers take all eight bits), thousands of cells with a
    frame or none, a ragged frame count (the last turn's second block is empty or partial).  Three passes with updates in
    between -- a full one over ungrouped frames, then incremental ones on the sorted list ("sorted": two blocks per turn;
    "auto": one, the shard is small) --, rows and codebooks against the oracle every time."""
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", accumulate)
    frames = _frames(20263, T, classes=11)
    refl = _codebook(oracle, frames, M, seed=12)
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        for it in range(3):
            cq = oracle.reflections_to_cq(refl)
            sym_o, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
            s.run_pass()
            assert s.last_pass_info()[0]
            kind, _two, _frac = s.last_pass_sweep()
            # (the first pass -- frames not grouped -- runs the unfused chain, or round 4's kernel where the records of a full
            # pass would not fit their bins: M = 8192; the incremental ones the fused kernel -- "auto": until its first sorted
            # pass has found most tiles flagged, as a codebook drawn from the frames at random makes them, then round 4's kernel)
            assert (kind == 3 or (accumulate == "auto" and it > 1 and kind == 1)) if it > 0 else kind in (1, 2), (it, kind)
            assert oracle_lib.rows_match(s.get_rows(), rows_o, P), f"pass {it}"
            refl, _failed = oracle.update(rows_o, P, sh_r, refl)
            s.update()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))


def test_two_stage_sweep_is_dropped_when_most_tiles_are_flagged(oracle, monkeypatch):
    """The host's switch (vq_pass.cpp: flagged share of the first pass behind a sort above 0.45 -> one stage for the rest of the
    level).  A codebook WITHOUT the ladder's tree order -- codewords drawn from the frames at random, so the codewords near a
    frame are scattered over the tiles -- flags most (tile, column block) jobs: pass 0 (frames not grouped) runs the unfused
    chain, pass 1 sorts, runs two stages and measures, passes 2.. run the fused kernel with ONE stage.  Symbols, distortions,
    rows and codebooks equal the oracle's on every pass, whichever sweep ran; a codebook grown by the ladder on the same
    frames keeps its two stages."""
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", "sorted")  # (the fused sorted pass from M = 64 on, two blocks per turn)
    T, M = 40947, 256  # (640 blocks: whole turns of two)
    frames = _frames(20271, T, classes=8)
    refl = _codebook(oracle, frames, M, seed=31)
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    sym, dmin = _DeviceBuffer(2 * T), _DeviceBuffer(8 * T)
    seen = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.sweep_executed(reset=True)
        for it in range(5):
            cq = oracle.reflections_to_cq(refl)
            sym_o, dmin_o, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
            s.run_pass(sym.ptr.value, dmin.ptr.value)
            s.pass_stats()
            kind, two, frac = s.last_pass_sweep()
            seen.append((kind, two, frac))
            assert np.array_equal(sym.to_host(np.uint16), sym_o), f"pass {it}"
            assert np.array_equal(dmin.to_host(np.uint64), dmin_o.view(np.uint64)), f"pass {it}"
            assert oracle_lib.rows_match(s.get_rows(), rows_o, P), f"pass {it}"
            refl, _failed = oracle.update(rows_o, P, sh_r, refl)
            s.update()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
        flagged, jobs, one_stage_jobs = s.sweep_executed()
    sym.free()
    dmin.free()
    assert seen[0][0] in (1, 2)                                   # frames not grouped yet
    assert seen[1][0] == 3 and seen[1][1] and seen[1][2] > 0.45   # sorted, two stages, most jobs flagged: measured
    assert all(k == 3 and not two for k, two, _f in seen[2:]), seen  # ... so the rest of the level runs one stage
    nblocks = (T + 63) // 64
    assert jobs == 2 * (M // 32) * nblocks and one_stage_jobs == 3 * jobs and flagged == round(seen[1][2] * jobs)
    # the product's accumulate (no ECOZ2_VQ_ACCUMULATE): once the sorted pass has measured that most jobs are flagged, the rest of
    # the level runs round 4's kernel -- frames in their natural order -- instead of a one-stage sorted pass; same bits
    monkeypatch.delenv("ECOZ2_VQ_ACCUMULATE")
    refl = _codebook(oracle, frames, M, seed=31)
    seen = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        for it in range(4):
            cq = oracle.reflections_to_cq(refl)
            _sym_o, _dmin_o, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
            s.run_pass()
            s.pass_stats()
            seen.append(s.last_pass_sweep()[:2] + (s.sweep_policy_state()[0],))
            assert oracle_lib.rows_match(s.get_rows(), rows_o, P), f"pass {it}"
            refl, _failed = oracle.update(rows_o, P, sh_r, refl)
            s.update()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
    assert seen[1][:2] == (3, True) and seen[1][2] >= M and all(k == 1 for k, _two, _u in seen[2:]), seen
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", "sorted")
    # the ladder's own codebook of the same size on the same frames: tree-ordered, two stages throughout
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, M // 2)
        s.grow()
        for it in range(3):
            s.run_pass()
            s.pass_stats()
            kind, two, frac = s.last_pass_sweep()
            assert kind == 3 and two and 0.0 <= frac < 0.45, (it, kind, two, frac)
            s.update()


@pytest.mark.parametrize("T", [20011, 65536])
def test_long_fallback_lists_take_the_wide_sweep(oracle, monkeypatch, T):
    """Data shaped like the reference's own corpus (r[0] = 1 / E about 2-3, no classes: e2vq_synth_frames_kind 1): the
    distortions are small differences of large terms and the three-limb keys leave 5-30 % of the frames uncertified.  From
    the second prefiltered pass on the host knows that and enqueues the FP64 fallback sweep in the plain pass's shape
    (k_pass_mfma<.., SRC = 3>: four frame tiles per wave); seeded and incremental passes, symbols, distortions, rows and
    codebooks against the oracle, then the whole ladder against oracle.learn."""
    frames = e.synth.synth_frames_kind(20281, 1, 6, 0.01, P, 0, T)
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    sym, dmin = _DeviceBuffer(2 * T), _DeviceBuffer(8 * T)
    shares = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.set_sweep_policy(-1.0, 1.0)  # (never the plain sweep: the prefiltered pass and its fallback are what is tested)
        s.learn(0.05, 128)
        for M in (256, 512):
            s.grow()
            refl = s.get_codebook()
            for it in range(3):
                cq = oracle.reflections_to_cq(refl)
                sym_o, dmin_o, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
                s.run_pass(sym.ptr.value, dmin.ptr.value)
                s.pass_stats()
                assert s.last_pass_info()[0]
                shares.append(s.sweep_policy_state()[2] / T)
                assert np.array_equal(sym.to_host(np.uint16), sym_o), (M, it)
                assert np.array_equal(dmin.to_host(np.uint64), dmin_o.view(np.uint64)), (M, it)
                assert oracle_lib.rows_match(s.get_rows(), rows_o, P), (M, it)
                refl, _failed = oracle.update(rows_o, P, sh_r, refl)
                s.update()
                assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
    sym.free()
    dmin.free()
    # (a pass behind one that left more than 2 % of the frames uncertified enqueues the wide sweep: that happened)
    assert any(x > 0.02 for x in shares[:-1]), shares
    # the whole ladder, switches at their defaults (the plain sweep may take over where more than 40 % stay uncertified)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, 1024)
    assert rc == 0
    cbs = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, 1024, callback=lambda *a: cbs.append(a))
        refl = s.get_codebook()
    assert [l.passes for l in levels] == [lv["passes"] for lv in levels_o] and cbs == cbs_o
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))


@pytest.mark.parametrize("few_div,expect", [("0", [True] * 5), ("1", [True, False, False, False, False])])
def test_few_records_switch_the_level_to_the_burst(oracle, monkeypatch, few_div, expect):
    """k_reduce_records publishes the pass's record count; below frames / ECOZ2_VQ_RECORDS_FEW_DIV the rest of the level adds
    its contributions as the burst of atomics inside the sweep (round 4).  Forced here (divisor 1: any incremental pass) and
    disabled (0): the two accumulates follow each other within a level and the rows equal the oracle's every pass."""
    monkeypatch.setenv("ECOZ2_VQ_RECORDS_FEW_DIV", few_div)
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", "records")  # (round 4's fused kernel: the sweeps of round 5 never switch)
    M = 256
    frames = _frames(20311, 9000)
    refl = np.concatenate([_codebook(oracle, frames, M // 2, seed=11)] * 2, axis=0)
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    recorded = []
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        for it in range(5):
            cq = oracle.reflections_to_cq(refl)
            _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
            s.run_pass()
            assert s.last_pass_info()[0]
            assert oracle_lib.rows_match(s.get_rows(), rows_o, P), f"pass {it}"
            s.pass_stats()
            recorded.append(s.last_pass_records()[0])
            refl, _failed = oracle.update(rows_o, P, sh_r, refl)
            s.update()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
        total = s.last_pass_records()[1]
    assert recorded == expect, recorded
    assert 0 <= total <= 2 * len(frames)


@pytest.mark.parametrize("split_max_m", [-2, -1, 0])
@pytest.mark.parametrize("collective", [False, True])
@pytest.mark.parametrize("min_m", [64, 256])
def test_seeded_first_pass_after_a_split_equals_a_full_accumulation(oracle, monkeypatch, min_m, collective, split_max_m):
    """The first pass of a level is seeded with the parents' exact sums (rows[2 i] = parent i, frames landing in 2 i + 1
    add once to a side table, only frames leaving their family are moved; k_seed_family / k_family_fixup): rows after that
    pass -- and after the incremental passes that build on it -- must equal the oracle's full accumulation bit for bit.
    Levels 32 -> 64 (parent on the plain sweep, which records the cells) up to 512, through grow / pass / update; with
    `collective` the parent rows come from the rank's own copy.
    split_max_m (the name is round 4's; its values now pick the kernels): -2 the fused pass over frames grouped by cell
    (round 5, here at every prefiltered size), whose in-block reduction adds to the side table and to the rows; -1 round 4's
    fused kernel with the recorded accumulate (seeded first passes whose records include the side table's bins, incremental
    ones after); 0 the same kernel with its burst of atomics."""
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", str(min_m))
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", {-2: "sorted", -1: "records", 0: "burst"}[split_max_m])
    monkeypatch.setenv("ECOZ2_VQ_QUIET", "1")
    if collective:
        monkeypatch.setenv("ECOZ2_VQ_FORCE_ALLREDUCE", "1")
    frames = _frames(20301, 20011, classes=7)
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    calls = []
    with e.VqSession(P) as s:
        if collective:
            s.set_allreduce(lambda buf, count, op, stream: calls.append(count) or 0, 0, 1)
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, 32)
        refl = s.get_codebook()
        seeded = 0
        for M in (64, 128, 256, 512):
            s.grow()
            refl = oracle.grow(refl) if hasattr(oracle, "grow") else s.get_codebook()
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
            for it in range(3):
                cq = oracle.reflections_to_cq(refl)
                _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
                s.run_pass()
                assert oracle_lib.rows_match(s.get_rows(), rows_o, P), f"M={M} pass {it}"
                if it < 2:
                    refl, _failed = oracle.update(rows_o, P, sh_r, refl)
                    s.update()
                    assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))
                else:
                    s.pass_stats()  # (the level ends on a pass without an update, as in e2vq_learn)
            seeded += M >= min_m
    assert seeded >= 2 and bool(calls) == collective


def test_save_and_restore_state_repeat_a_level(oracle, monkeypatch):
    """e2vq_save_state / e2vq_restore_state: the point where a level ended (codebook, DDprv, rows, cells) comes back,
    and the next level -- seeded first pass included -- repeats bit for bit, as often as asked (bench.py's timed region)."""
    monkeypatch.setenv("ECOZ2_VQ_QUIET", "1")
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    frames = _frames(20311, 15000, classes=5)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, 128)
        s.save_state()
        first = s.learn(0.05, 256)[0]
        cb = s.get_codebook()
        for _ in range(3):
            s.restore_state()
            again = s.learn(0.05, 256)[0]
            assert (again.passes, again.DD, again.sigma, again.inertia) == (first.passes, first.DD, first.sigma, first.inertia)
            assert np.array_equal(s.get_codebook().view(np.uint64), cb.view(np.uint64))
    rc, levels_o, _ = oracle.learn(frames, 0.05, 256)
    assert np.array_equal(cb.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))


def test_incremental_switch_gives_the_same_ladder(oracle, monkeypatch):
    """ECOZ2_VQ_ACCUMULATE=full (full accumulation every pass) and the default produce identical ladders."""
    frames = _frames(20262, 12000, classes=5)
    out = []
    for acc in ("auto", "full"):
        monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", acc)
        cbs = []
        with e.VqSession(P) as s:
            s.set_frames(frames)
            s.prepare()
            s.init_codebook()
            s.learn(0.05, 512, callback=lambda M, a, sg, i: cbs.append((M, a, sg, i)))
            out.append((cbs, s.get_codebook().copy()))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1].view(np.uint64), out[1][1].view(np.uint64))


def test_iterate_is_pass_stats_update(oracle, monkeypatch):
    """e2vq_iterate = e2vq_pass + e2vq_pass_stats + e2vq_update in one call (what bench.py times)."""
    frames = _frames(20263, 7000)
    refl = _codebook(oracle, frames, 64, seed=11)
    rc, st = oracle.data_stats(frames)
    sh_r, sh_q = oracle.shifts(st.maxabs)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        for _ in range(3):
            cq = oracle.reflections_to_cq(refl)
            Ed = oracle.dist_exponent(cq, st.maxabs)
            _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, Ed)
            ls_o = oracle.rows_stats(rows_o, P, frames.shape[0], sh_r, Ed, oracle.unfix(st.q_hi, st.q_lo, sh_q))
            refl, _failed = oracle.update(rows_o, P, sh_r, refl)
            ls = s.iterate()
            assert (ls.DD, ls.avg_distortion, ls.sigma, ls.inertia) == (ls_o.DD, ls_o.avg, ls_o.sigma, ls_o.inertia)
            assert np.array_equal(s.get_codebook().view(np.uint64), refl.view(np.uint64))


@pytest.mark.parametrize("Pn", [12, 16, 20, 24, 28, 32, 40])
def test_prefilter_for_other_prediction_orders(oracle, monkeypatch, Pn):
    """the K-slot packing is generic in the prediction order (P = 12, 16, ..., 40; `-P` is a free parameter of the
    reference's CLI, src/vq/mod.rs:44): whole ladders to M = 512 with the prefiltered sweep from M = 64 (full and
    incremental accumulates, fallback list), a quantize through the prefiltered quantize path, and one pass with
    adversarial frames -- all bit-identical to the oracle"""
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER", "1")
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    frames = e.synth.synth_frames(20270 + Pn, 6, Pn, 0, 12000)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, 512)
    assert rc == 0
    cbs, used = [], []
    with e.VqSession(Pn) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, 512, callback=lambda M, a, sg, i: (cbs.append((M, a, sg, i)), used.append(s.last_pass_info())))
        refl = s.get_codebook()
        sym, dmin = s.quantize(frames)
        # adversarial pass: rescaled / zero / sign-flipped frames, duplicated codewords
        rng = np.random.default_rng(Pn)
        bad = frames[:3000].copy()
        bad[::7] *= 10.0 ** rng.integers(-15, 15, size=bad[::7].shape[0])[:, None]
        bad[5::11] = 0.0
        bad[3::13] *= -1.0
        twins = np.concatenate([refl[:128], refl[:128]], axis=0)
        s.set_frames(bad)
        s.prepare()
        s.set_codebook(twins)
        s.run_pass()
        rows_bad = s.get_rows()
        pre_bad, fb_bad = s.last_pass_info()
    assert [(l.M, l.passes) for l in levels] == [(l["M"], l["passes"]) for l in levels_o] and cbs == cbs_o
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    assert [u[0] for u in used] == [lv["M"] >= 64 for lv in levels_o]  # the prefiltered kernel did serve M >= 64
    assert all(u[1] < 0.2 * len(frames) for u in used)
    sym_o, dmin_o = oracle.quantize(oracle.reflections_to_cq(refl), frames)
    assert np.array_equal(sym, sym_o) and np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))
    cq = oracle.reflections_to_cq(twins)
    rc, st = oracle.data_stats(bad)
    sh_r, _ = oracle.shifts(st.maxabs)
    _s, _d, rows_o = oracle.run_pass(cq, bad, sh_r, oracle.dist_exponent(cq, st.maxabs))
    assert pre_bad and fb_bad > 0, (pre_bad, fb_bad)
    assert oracle_lib.rows_match(rows_bad, rows_o, Pn), np.argwhere(rows_bad != rows_o)[:10]


@pytest.mark.parametrize("records", ["1", "0"])
def test_order_40_runs_seven_waves_per_workgroup(oracle, monkeypatch, records):
    """P = 40: 21 KB of FP64 frames per wave -- seven waves per workgroup instead of eight in the LDS-staged pass and in
    fused quantize (round 4) -- and rows of 83 elements, which the burst of atomics cannot add: with ECOZ2_VQ_ACCUMULATE=burst the
    training passes run the plain FP64 sweep (round 2's accumulating kernel, which used to serve that case, left in round 5).
    Default thresholds (prefilter and seeding from M = 128, the round-5 kernels from 256), ladder to 1024 on a ragged frame
    count."""
    monkeypatch.setenv("ECOZ2_VQ_ACCUMULATE", "auto" if records == "1" else "burst")
    Pn = 40
    frames = e.synth.synth_frames(20340, 9, Pn, 0, 64 * 7 * 9 + 37)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, 1024)
    assert rc == 0
    cbs = []
    with e.VqSession(Pn) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, 1024, callback=lambda M, a, sg, i: cbs.append((M, a, sg, i)))
        refl = s.get_codebook()
        used = s.last_pass_info()[0]
        sym, dmin = s.quantize(frames)
    assert used == (records == "1")
    assert [(l.M, l.passes) for l in levels] == [(l["M"], l["passes"]) for l in levels_o] and cbs == cbs_o
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    sym_o, dmin_o = oracle.quantize(oracle.reflections_to_cq(refl), frames)
    assert np.array_equal(sym, sym_o) and np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("first,T", [(1, 4097), (3, 64), (5, 63), (2, 1), (7, 12345)])
def test_device_resident_quantize_at_unaligned_offsets(oracle, monkeypatch, fused, first, T):
    """e2vq_quantize_device on a slice of a larger device buffer: the slice starts at an odd frame (37 doubles per frame:
    only 8-byte aligned), sizes ragged around the 64-frame blocks.  The fused kernel stages whole blocks with 16-byte
    LDS-DMA loads and the last, partial block element by element; both must give the oracle's symbols and distortions."""
    if not fused:
        monkeypatch.setenv("ECOZ2_VQ_QUANTIZE_UNFUSED", "1")
    frames = _frames(20260, first + T + 3)
    refl = _codebook(oracle, frames[: max(2000, T)] if T > 2000 else _frames(20261, 4000), 256, seed=9)
    buf = _DeviceBuffer(frames.nbytes)
    buf.from_host(frames)
    sym, dmin = _DeviceBuffer(2 * T + 2), _DeviceBuffer(8 * T)
    with e.VqSession(P) as s:
        s.set_frames(_frames(20262, 3000))  # (any training set: the session only needs its scale for the prefilter)
        s.prepare()
        s.set_codebook(refl)
        s.quantize_device(buf.ptr.value + first * (P + 1) * 8, T, sym.ptr.value, dmin.ptr.value)
        s.synchronize()
    sym_o, dmin_o = oracle.quantize(oracle.reflections_to_cq(refl), frames[first : first + T])
    assert np.array_equal(sym.to_host(np.uint16)[:T], sym_o)
    assert np.array_equal(dmin.to_host(np.float64).view(np.uint64), dmin_o.view(np.uint64))
    for b in (buf, sym, dmin):
        b.free()


@pytest.mark.parametrize("M", [64, 256])
def test_statistics_need_a_pass_on_the_current_codebook(oracle, monkeypatch, M):
    """The distortion sums in the rows are fixed-point numbers scaled for the codebook the pass ran on: once e2vq_update
    has committed a new codebook, e2vq_pass_stats / e2vq_update must refuse to work from the stale rows (they used to
    recompute statistics with the wrong scale), and the next pass must give the oracle's rows again."""
    monkeypatch.setenv("ECOZ2_VQ_PREFILTER_MIN_M", "64")
    frames = _frames(20264, 9000)
    refl = _codebook(oracle, frames, M, seed=12)
    rc, st = oracle.data_stats(frames)
    sh_r, sh_q = oracle.shifts(st.maxabs)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        with pytest.raises(RuntimeError, match="has not run"):
            s.pass_stats()  # nothing has run yet
        s.run_pass()
        ls1 = s.pass_stats()
        assert s.pass_stats().DD == ls1.DD  # (cached)
        s.update()
        with pytest.raises(RuntimeError, match="has not run"):
            s.pass_stats()
        with pytest.raises(RuntimeError, match="has not run"):
            s.update()
        cq = oracle.reflections_to_cq(oracle.update(_oracle_pass(oracle, frames, refl)[2], P, sh_r, refl)[0])
        _sym, _dmin, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
        s.run_pass()
        assert oracle_lib.rows_match(s.get_rows(), rows_o, P)
