"""GPU tests through the reference's own entry points (C-ABI part 1) and at the BASELINE sizes."""
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import ecoz2rs_amd as e
from tests import oracle_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
P = 36


def test_ecoz2_vq_learn_and_quantize_reproduce_golden_files(tmp_path, monkeypatch, oracle):
    """config 1 through ecoz2_vq_learn / ecoz2_vq_quantize: emitted .cbook / .seq are byte-identical to the fixtures."""
    meta = json.load(open(os.path.join(GOLD, "config1.json")))
    frames = e.synth.synth_frames(meta["seed"], meta["classes"], P, 0, meta["T"])
    # several .prd files of ragged length; sorted file order is the frame order (src/utl/mod.rs:216-218)
    cuts = [0, 1, 1000, 1777, 4096, 9999, 10000]
    files = []
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        f = tmp_path / "data" / "predictors" / "_" / f"{i:05d}.prd"
        e.formats.write_prd(str(f), "_", frames[a:b])
        files.append(str(f))
    monkeypatch.setenv("ECOZ2_VQ_OUT_ROOT", str(tmp_path))
    monkeypatch.setenv("ECOZ2_VQ_MAX_CODEBOOK_SIZE", str(meta["max_M"]))
    seen = []
    e.vq_learn(None, P, meta["eps"], "_", files, callback=lambda *a: seen.append(a))
    assert [s[0] for s in seen] == [g["M"] for g in meta["levels"]]
    for s, g in zip(seen, meta["levels"]):
        assert (s[1].hex(), s[2].hex(), s[3].hex()) == (g["avg"], g["sigma"], g["inertia"])
        name = f"eps_0.05_M_{g['M']:04d}.cbook"
        got = open(tmp_path / "data" / "codebooks" / "_" / name, "rb").read()
        assert got == open(os.path.join(GOLD, "config1_" + name), "rb").read()
    assert (tmp_path / "data" / "codebooks" / "_" / "eps_0.05.rpt").exists()
    # quantize one .prd holding all frames against the M=16 codebook
    whole = tmp_path / "data" / "predictors" / "_" / "all.prd"
    e.formats.write_prd(str(whole), "_", frames)
    e.vq_quantize(str(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_0016.cbook"), [str(whole)], True)
    got = open(tmp_path / "data" / "sequences" / "M16" / "_" / "all.seq", "rb").read()
    assert got == open(os.path.join(GOLD, "config1_M0016.seq"), "rb").read()
    # resume from the M=4 codebook (-B): trains 8 and 16 (CHANGELOG.md:366-368)
    seen2 = []
    e.vq_learn(str(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_0004.cbook"), None, 1e9, "_", files,
               callback=lambda *a: seen2.append(a))
    assert [s[0] for s in seen2] == [8, 16]
    # ... and the resumed codebooks / callback scalars are the oracle's for the same base (F1b,
    # src/ecoz2_lib/mod.rs:107-115): bytes of eps_1e+09_M_0008/0016.cbook against oracle.learn(base=M4)
    _c, _p, base = e.formats.read_cbook(os.path.join(GOLD, "config1_eps_0.05_M_0004.cbook"))
    rc, levels_b, cbs_b = oracle.learn(frames, 1e9, meta["max_M"], base=base)
    assert rc == 0 and [lv["M"] for lv in levels_b] == [8, 16] and seen2 == cbs_b
    for lv in levels_b:
        got = open(tmp_path / "data" / "codebooks" / "_" / f"eps_1e+09_M_{lv['M']:04d}.cbook", "rb").read()
        want = tmp_path / "want.cbook"
        e.formats.write_cbook(str(want), "_", lv["reflections"])
        assert got == open(want, "rb").read()
    # a resume with the real epsilon (several passes per level) as well
    seen3 = []
    e.vq_learn(str(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_0002.cbook"), None, 0.01, "_", files,
               callback=lambda *a: seen3.append(a))
    _c, _p, base2 = e.formats.read_cbook(os.path.join(GOLD, "config1_eps_0.05_M_0002.cbook"))
    rc, levels_c, cbs_c = oracle.learn(frames, 0.01, meta["max_M"], base=base2)
    assert rc == 0 and seen3 == cbs_c and [lv["M"] for lv in levels_c] == [4, 8, 16]
    for lv in levels_c:
        _c, _p, refl = e.formats.read_cbook(str(tmp_path / "data" / "codebooks" / "_" / f"eps_0.01_M_{lv['M']:04d}.cbook"))
        assert np.array_equal(refl.view(np.uint64), lv["reflections"].view(np.uint64))


@pytest.mark.parametrize("T", [1, 15, 16, 17, 63, 64, 65, 129, 4097])
def test_ragged_sizes(oracle, T):
    """Block / tile padding never leaks: any T gives the oracle's rows, symbols and codebook."""
    frames = e.synth.synth_frames(31, 3, P, 5, T)
    refl = np.zeros((5, P + 1))  # M = 5: not a multiple of the 16-codeword MFMA tile
    for i in range(5):
        refl[i, 1:] = oracle.lpca_r(e.synth.synth_frames(32, 3, P, i, 1)[0], P)[2][1:]
    cq = oracle.reflections_to_cq(refl)
    rc, st = oracle.data_stats(frames)
    sh_r, _ = oracle.shifts(st.maxabs)
    sym_o, dmin_o, rows_o = oracle.run_pass(cq, frames, sh_r, oracle.dist_exponent(cq, st.maxabs))
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.run_pass()
        assert oracle_lib.rows_match(s.get_rows(), rows_o, P)
        sym, dmin = s.quantize(frames)
    assert np.array_equal(sym, sym_o) and np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))


def test_invalid_inputs_fail_loudly():
    frames = e.synth.synth_frames(1, 2, P, 0, 100)
    with e.VqSession(P) as s:
        with pytest.raises(e.Ecoz2Error):
            s.run_pass()  # no training set / codebook
        bad = frames.copy()
        bad[7, 3] = np.inf
        s.set_frames(bad)
        with pytest.raises(e.Ecoz2Error, match="NaN or infinite"):
            s.prepare()
        s.set_frames(np.zeros((10, P + 1)))
        with pytest.raises(e.Ecoz2Error, match="all zeros"):
            s.prepare()
    with pytest.raises(e.Ecoz2Error):
        e.vq_quantize("/nonexistent.cbook", [])


@pytest.mark.parametrize("Pn", [4, 7, 10, 12, 16, 20, 24, 30, 33, 35, 38, 40, 41, 44, 48, 63, 64, 65, 79, 80, 81])
def test_other_prediction_orders(oracle, Pn):
    """Every P = 4..80 runs on the matrix pipe (NC = 4k+1: trailing coefficient on the VALU; otherwise a zero-padded
    last k-step with 2, 3 or 4 live coefficients; P > 40, round 4: a wave sweeps a 32-frame half block, the accumulate goes
    straight to global atomics, P > 63 with the thread-per-cell update kernels); larger orders (81 here) use the LDS-staged
    VALU kernel."""
    frames = e.synth.synth_frames(41, 3, Pn, 0, 3000)
    max_m = 256 if Pn in (7, 10, 12, 24, 30, 33, 35, 38, 40, 48, 64) else 8  # 256: LDS-table and hybrid accumulate modes
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, max_m)
    assert rc == 0
    cbs = []
    with e.VqSession(Pn) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, max_m, callback=lambda *a: cbs.append(a))
        refl = s.get_codebook()
        sym, dmin = s.quantize(frames)
    assert cbs == cbs_o
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    sym_o, dmin_o = oracle.quantize(oracle.reflections_to_cq(refl), frames)
    assert np.array_equal(sym, sym_o) and np.array_equal(dmin.view(np.uint64), dmin_o.view(np.uint64))


def test_config2_sized_learn_pass_against_oracle(oracle):
    """configs[1] scale (1M frames, M=256): one full LBG iteration, bit-exact against the oracle (a few seconds of CPU)."""
    T, M = 1 << 20, 256
    frames = e.synth.synth_frames(20242, 20, P, 0, T)
    refl = np.zeros((M, P + 1))
    for i in range(M):
        refl[i, 1:] = oracle.lpca_r(frames[i * 4001], P)[2][1:]
    cq = oracle.reflections_to_cq(refl)
    rc, st = oracle.data_stats(frames)
    sh_r, sh_q = oracle.shifts(st.maxabs)
    Ed = oracle.dist_exponent(cq, st.maxabs)
    _so, _do, rows_o = oracle.run_pass(cq, frames, sh_r, Ed)
    refl_o, _ = oracle.update(rows_o, P, sh_r, refl)
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.set_codebook(refl)
        s.run_pass()
        rows = s.get_rows()
        s.update()
        refl_g = s.get_codebook()
    assert oracle_lib.rows_match(rows, rows_o, P)
    assert np.array_equal(refl_g.view(np.uint64), refl_o.view(np.uint64))
    assert int(rows[:, 2 * (P + 1)].sum()) == T  # every frame counted exactly once


def test_config2_full_ladder_cbook_bit_exact(oracle, tmp_path):
    """configs[1] as BASELINE.json words it: vq learn on 1M frames, P = 36, up to M = 256 -- every level's pass count and
    callback scalars, and the bytes of the final .cbook, equal the oracle's (whole ladder, ~30 s of CPU for the oracle)."""
    T, M = 1_000_000, 256
    frames = e.synth.synth_frames(20242, 20, P, 0, T)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, M, out_root=str(tmp_path / "o"))
    assert rc == 0
    cbs = []
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, M, out_root=str(tmp_path / "g"), callback=lambda *a: cbs.append(a))
    assert [(l.M, l.passes) for l in levels] == [(l["M"], l["passes"]) for l in levels_o] and cbs == cbs_o
    for lv in levels_o:
        name = os.path.join("data", "codebooks", "_", f"eps_0.05_M_{lv['M']:04d}.cbook")
        assert open(tmp_path / "g" / name, "rb").read() == open(tmp_path / "o" / name, "rb").read()


def test_config3_sized_quantize_properties(oracle, tmp_path, monkeypatch):
    """configs[2] scale: 10M frames against M=1024.  Full oracle check on a 2M-frame slice; size-independent
    properties on all 10M: chunking invariance, dmin recomputed independently, every symbol < M.  Then the same 10 M
    frames as ONE .prd through ecoz2_vq_quantize with 1, 2 and 5 workers (the file is split into chunks that any
    worker takes; the .seq is written piecewise): byte-identical .seq whatever the worker count."""
    T, M = 10_000_000, 1024
    frames = e.synth.synth_frames(20243, 20, P, 0, T)
    refl = np.zeros((M, P + 1))
    for i in range(M):
        refl[i, 1:] = oracle.lpca_r(frames[i * 9001], P)[2][1:]
    cq = oracle.reflections_to_cq(refl)
    with e.VqSession(P) as s:
        s.set_codebook(refl)
        sym, dmin = s.quantize(frames)
        sym2 = s.quantize(frames[1234567:1234567 + 300001], want_dmin=False)  # different chunk boundaries
    assert sym.max() < M and np.array_equal(sym2, sym[1234567:1234567 + 300001])
    sl = slice(3_000_000, 5_000_000)
    sym_o, dmin_o = oracle.quantize(cq, frames[sl])
    assert np.array_equal(sym[sl], sym_o) and np.array_equal(dmin[sl].view(np.uint64), dmin_o.view(np.uint64))
    # the reported minimum is the distortion of the reported codeword (independent fp64 evaluation, sampled)
    idx = np.random.default_rng(0).choice(T, 20000, replace=False)
    d = np.einsum("ij,ij->i", frames[idx], cq[sym[idx]])
    assert np.allclose(d, dmin[idx], rtol=1e-12, atol=1e-12)
    # one 10 M-frame file (2.96 GB), split over the workers
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    prd = tmp_path / "data" / "predictors" / "_" / "all.prd"
    cb = tmp_path / "cb.cbook"
    e.formats.write_prd(str(prd), "_", frames)
    e.formats.write_cbook(str(cb), "_", refl)
    del frames
    monkeypatch.setenv("ECOZ2_VQ_OUT_ROOT", str(tmp_path))
    want = None
    for workers in (1, 2, 5):
        monkeypatch.setenv("ECOZ2_VQ_GPUS", str(workers))
        seq = tmp_path / "data" / "sequences" / f"M{M}" / "_" / "all.seq"
        if seq.exists():
            seq.unlink()
        e.vq_quantize(str(cb), [str(prd)])
        got = open(seq, "rb").read()
        if want is None:
            want = got
            cls, m, sy = e.formats.read_seq(str(seq))
            assert (cls, m) == ("_", M) and np.array_equal(sy, sym)  # = the session call's symbols (oracle-checked slice)
        assert got == want, f"{workers} workers"


def test_config4_sized_learn_properties(oracle):
    """configs[3]'s data size on ONE GPU: 2^24 frames, whole LBG ladder 2..1024 with the real convergence rule, then one
    more pass at M = 1024 whose outputs are checked through size-independent properties and sampled oracle runs:
      * every frame counted once; per-cell counts == histogram of the emitted symbols; DD == sum of (dmin - 1);
      * a contiguous 400k-frame slice: symbols and min distortions bit-equal to the oracle;
      * all member frames of 24 sampled cells (~400k frames picked by the GPU's own symbols): the oracle assigns them to
        the same cells and its exact integer cell sums equal the GPU's rows of those cells word for word -- the
        accumulate of the full 16M-frame pass, checked exactly where it was sampled;
      * shard invariance: the first and second half of the set swept as separate quantize calls (different block /
        chunk boundaries) give the same symbols.
    (No torch in this process: torch bundles its own HIP/HSA runtime, and a second runtime initialised after the
    library's finds no device.  The symbols of the training pass are tied to the quantize entry's through the cell
    counts, which must equal the histogram of the quantize symbols.)"""
    T, M = 1 << 24, 1024
    frames = e.synth.synth_frames(20244, 20, P, 0, T)
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, M)
        assert [l.M for l in levels] == [2 << i for i in range(10)] and all(l.passes >= 2 for l in levels)
        assert all(a.avg_distortion > b.avg_distortion for a, b in zip(levels, levels[1:]))
        refl = s.get_codebook()
        s.run_pass()
        st = s.pass_stats()
        rows = s.get_rows()
        prefiltered, fallback = s.last_pass_info()
        # the same assignment through the quantize entry (other kernels, other chunk boundaries: 4M-frame chunks)
        sym, dmin = s.quantize(frames)
        half = T // 2 + 12345
        sym_a = s.quantize(frames[:half], want_dmin=False)
        sym_b = s.quantize(frames[half:], want_dmin=False)
    assert prefiltered and 0 <= fallback < T // 20
    assert np.array_equal(sym[:half], sym_a) and np.array_equal(sym[half:], sym_b)
    counts = rows[:, 2 * (P + 1)]
    assert int(counts.sum()) == T and np.array_equal(counts, np.bincount(sym, minlength=M))
    assert st.empty_cells == int((counts == 0).sum())
    import math
    dd = math.fsum((dmin - 1.0).tolist())
    assert abs(st.DD - dd) <= 1e-9 * abs(dd) and st.avg_distortion == st.DD / T
    # sampled oracle checks
    cq = oracle.reflections_to_cq(refl)
    sl = slice(9_000_000, 9_400_000)
    sym_o, dmin_o = oracle.quantize(cq, frames[sl])
    assert np.array_equal(sym[sl], sym_o) and np.array_equal(dmin[sl].view(np.uint64), dmin_o.view(np.uint64))
    maxabs = float(np.abs(frames).max())
    sh_r, _ = oracle.shifts(maxabs)
    Ed = oracle.dist_exponent(cq, maxabs)
    order = np.argsort(counts)
    cells = np.concatenate([order[-8:], order[M // 2 - 8:M // 2 + 8]])  # the 8 fullest cells and 16 median ones
    cells = cells[counts[cells] > 0]
    idx = np.flatnonzero(np.isin(sym, cells))
    assert len(idx) >= 200_000
    sym_m, dmin_m, rows_m = oracle.run_pass(cq, frames[idx], sh_r, Ed)
    assert np.array_equal(sym_m, sym[idx]) and np.array_equal(dmin_m.view(np.uint64), dmin[idx].view(np.uint64))
    n_lc = 2 * (P + 1) + 1  # limb sums and count (the distortion columns are defined by their totals only)
    assert np.array_equal(rows_m[cells][:, :n_lc], rows[cells][:, :n_lc])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import ecoz2rs_amd as e
from ecoz2rs_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
T, MAXM, SEED = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
lo, hi = parallel.shard_range(T, rank, world)
frames = e.synth.synth_frames(SEED, 5, 36, lo, hi - lo)
os.environ["ECOZ2_VQ_QUIET"] = "1"
s = e.VqSession(36, device=0)
parallel.bind_torch_stream(s, 0)
s.set_allreduce(parallel.make_allreduce(0), rank, world)
s.set_frames(frames); s.prepare(); s.init_codebook()
levels = s.learn(0.05, MAXM)
np.save(sys.argv[2] + f"/cb_{rank}.npy", s.get_codebook())
np.save(sys.argv[2] + f"/passes_{rank}.npy", np.array([l.passes for l in levels]))
s.close(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("T,max_m,seed,world", [(50001, 512, 55, 2), (100003, 1024, 57, 2), (100003, 1024, 57, 3)])
def test_two_ranks_on_one_gpu_equal_single_rank(tmp_path, oracle, T, max_m, seed, world):
    """Sharded learn (N processes, gloo exchange of the int64 cell sums) gives the single-rank codebook bit-for-bit --
    up to config 4's codebook size M = 1024 (prefiltered sweep + incremental accumulate under a real exchange) --
    and both equal the oracle's ladder."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), str(T), str(max_m), str(seed)],
                                      env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    frames = e.synth.synth_frames(seed, 5, P, 0, T)
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, max_m)
        ref = s.get_codebook()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"cb_{r}.npy").view(np.uint64), ref.view(np.uint64))
        assert list(np.load(tmp_path / f"passes_{r}.npy")) == [l.passes for l in levels]
    rc, levels_o, _cbs = oracle.learn(frames, 0.05, max_m)
    assert rc == 0 and [l.passes for l in levels] == [lv["passes"] for lv in levels_o]
    assert np.array_equal(ref.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    assert [l.DD for l in levels] == [lv["DD"] for lv in levels_o]


_NCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import ecoz2rs_amd as e
from ecoz2rs_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
frames = e.synth.synth_frames(56, 5, 36, 0, 20000)
os.environ["ECOZ2_VQ_QUIET"] = "1"
os.environ["ECOZ2_VQ_FORCE_ALLREDUCE"] = "1"
calls = []
hook = parallel.make_allreduce(0)
def counting(ptr, count, op, stream):
    calls.append((count, op)); hook(ptr, count, op, stream)
s = e.VqSession(36, device=0)
parallel.bind_torch_stream(s, 0)
s.set_allreduce(counting, 0, 1)
s.set_frames(frames); s.prepare(); s.init_codebook()
s.learn(0.05, 256)
np.save(sys.argv[2] + "/cb_nccl.npy", s.get_codebook())
np.save(sys.argv[2] + "/calls.npy", np.array(calls))
s.close(); dist.barrier(); dist.destroy_process_group()
"""


def test_rccl_hook_single_rank_group(tmp_path):
    """The exact `nccl` (RCCL) exchange code of bench.py, on a 1-rank group: int64 SUM/MAX all-reduce of the
    session's device buffers on the bound torch stream must leave the result unchanged."""
    script = tmp_path / "nccl.py"
    script.write_text(_NCCL_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], env=env, timeout=600).returncode == 0
    frames = e.synth.synth_frames(56, 5, P, 0, 20000)
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, 256)
        ref = s.get_codebook()
    assert np.array_equal(np.load(tmp_path / "cb_nccl.npy").view(np.uint64), ref.view(np.uint64))
    calls = np.load(tmp_path / "calls.npy")
    # e2vq_prepare: ONE maximum of two words -- max|x| and the bad-data flags (a NaN in any shard stops every rank) --, then the
    # SUM of the data statistics (which need the scale that maximum defines)
    assert calls[0].tolist() == [2, 1] and calls[1].tolist() == [2 * 37 + 3, 0]
    assert all(c[1] == 0 and c[0] % e.lib.e2vq_row_stride(P) == 0 for c in calls[2:])  # per-pass row all-reduces


def test_vq_classify(tmp_path, oracle, capfd):
    """ecoz2_vq_classify (src/ecoz2_lib/mod.rs:124-130): min average distortion over class codebooks."""
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    classes = {"A": 101, "Bm": 202, "C3": 303}
    cb_files, prd_files, truth = [], [], []
    for cls, seed in classes.items():
        train = e.synth.synth_frames(seed, 1, P, 0, 4000)
        with e.VqSession(P) as s:
            s.set_frames(train)
            s.prepare()
            s.init_codebook()
            s.learn(0.05, 8)
            refl = s.get_codebook()
            # average distortion through the C-ABI == oracle's min distortions summed in frame order
            test = e.synth.synth_frames(seed, 1, P, 10_000, 700)
            avg = s.avg_distortion(test)
        _sym, dmin = oracle.quantize(oracle.reflections_to_cq(refl), test)
        acc = 0.0
        for d in dmin:
            acc += d - 1.0
        assert avg == acc / len(dmin)
        cb = tmp_path / "data" / "codebooks" / cls / "eps_0.05_M_0008.cbook"
        e.formats.write_cbook(str(cb), cls, refl)
        cb_files.append(str(cb))
        for i in range(3):
            f = tmp_path / "data" / "predictors" / cls / f"{i:05d}.prd"
            e.formats.write_prd(str(f), cls, e.synth.synth_frames(seed, 1, P, 20_000 + 500 * i, 300 + 17 * i))
            prd_files.append(str(f))
    capfd.readouterr()
    e.vq_classify(cb_files, prd_files, show_ranked=True)
    out = capfd.readouterr().out
    assert "TOTAL" in out and "100.00%" in out.split("TOTAL")[1]
    for cls in classes:
        assert any(line.startswith(cls) and "100.00%" in line for line in out.splitlines())
    # a deliberately mislabeled file is reported with its ranking
    bad = tmp_path / "data" / "predictors" / "A" / "bad.prd"
    e.formats.write_prd(str(bad), "A", e.synth.synth_frames(classes["C3"], 1, P, 30_000, 200))
    e.vq_classify(cb_files, prd_files + [str(bad)], show_ranked=True)
    out = capfd.readouterr().out
    assert "classified as 'C3'; ranked: C3(" in out and "90.00%" in out.split("TOTAL")[1]
    # the same through 1 024-frame units (files batched and split: the streaming path), same report
    os.environ["ECOZ2_VQ_QUANTIZE_CHUNK"] = "1024"
    try:
        e.vq_classify(cb_files, prd_files + [str(bad)], show_ranked=True)
    finally:
        del os.environ["ECOZ2_VQ_QUANTIZE_CHUNK"]
    assert capfd.readouterr().out == out


def test_large_prediction_order_generic_path(oracle):
    """P = 70 (> 63): generic sweep kernel + thread-per-cell K3/K4 kernels (no wave-per-cell fusion)."""
    Pn = 70
    frames = e.synth.synth_frames(43, 2, Pn, 0, 1500)
    rc, levels_o, cbs_o = oracle.learn(frames, 0.05, 4)
    assert rc == 0
    cbs = []
    with e.VqSession(Pn) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        s.learn(0.05, 4, callback=lambda *a: cbs.append(a))
        refl = s.get_codebook()
    assert cbs == cbs_o
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))


def test_cli_end_to_end(tmp_path, oracle):
    """`ecoz2 vq learn` -> `vq quantize` -> `vq classify` -> `seq show` through the CLI binary (reference flags)."""
    exe = os.path.join(ROOT, "ecoz2rs_amd", "csrc", "ecoz2")
    env = dict(os.environ, ECOZ2_VQ_MAX_CODEBOOK_SIZE="8")
    env.pop("ECOZ2_VQ_OUT_ROOT", None)
    env.pop("ECOZ2_VQ_QUIET", None)

    def run(*args):
        r = subprocess.run([exe, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return r.stdout

    classes = {"A": 11, "B": 22}
    rows = ["tt,class,selection"]
    for cls, seed in classes.items():
        for i in range(4):
            fr = e.synth.synth_frames(seed, 1, P, 1000 * i, 400)
            e.formats.write_prd(str(tmp_path / "data" / "predictors" / cls / f"{i:05d}.prd"), cls, fr)
            rows.append(f"{'TRAIN' if i < 3 else 'TEST'},{cls},{i:05d}")
    (tmp_path / "tt.csv").write_text("\n".join(rows) + "\n")
    for cls in classes:  # one codebook per class from its TRAIN files (tt-list + --class-name, src/vq/mod.rs:50-58)
        out = run("vq", "learn", "-P", "36", "-e", "0.05", "--class-name", cls, "--predictors", "tt.csv")
        assert "predictor_filenames: 3" in out and f"codebook_class_name={cls}" in out
        assert "Ecoz2ObserverRef.step: M=8" in out and "1200 training vectors" in out
        assert (tmp_path / "data" / "codebooks" / cls / "eps_0.05_M_0008.cbook").exists()
    # the codebook the CLI wrote equals the oracle's for the same frames in sorted-file order
    frames_a = np.concatenate([e.formats.read_prd(str(tmp_path / "data" / "predictors" / "A" / f"{i:05d}.prd"))[2]
                               for i in range(3)])
    rc, levels_o, _ = oracle.learn(frames_a, 0.05, 8)
    _c, _p, refl = e.formats.read_cbook(str(tmp_path / "data" / "codebooks" / "A" / "eps_0.05_M_0008.cbook"))
    assert np.array_equal(refl.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
    out = run("vq", "quantize", "--codebook", "data/codebooks/A/eps_0.05_M_0008.cbook", "--predictors",
              "data/predictors", "-s")
    assert "number of predictor files: 8" in out
    seq = tmp_path / "data" / "sequences" / "M8" / "B" / "00003.seq"
    assert seq.exists()
    out = run("seq", "show", str(seq))
    assert out.startswith("<B(M=8,L=400): ")
    out = run("vq", "classify", "--codebooks", "data/codebooks/A/eps_0.05_M_0008.cbook",
              "data/codebooks/B/eps_0.05_M_0008.cbook", "--tt", "TEST", "--predictors", "tt.csv")
    assert "number of codebooks: 2  number of predictors: 2" in out
    assert "100.00%" in out.split("TOTAL")[1]
    # resume from a base codebook with the CLI (-B): next size only
    env["ECOZ2_VQ_MAX_CODEBOOK_SIZE"] = "16"
    out = run("vq", "learn", "-B", "data/codebooks/A/eps_0.05_M_0008.cbook", "--predictors", "data/predictors/A")
    assert "Ecoz2ObserverRef.step: M=16" in out and "M=8 " not in out.split("base codebook")[1]


@pytest.mark.parametrize("ranks", [2, 3, 5])
def test_in_process_group_reproduces_golden_codebooks(tmp_path, monkeypatch, ranks):
    """ECOZ2_VQ_GPUS=N: ecoz2_vq_learn shards the frames over N in-process ranks (sharing the one GPU here) and
    exchanges the int64 cell sums device-to-device; the codebooks must equal the single-rank / oracle fixtures."""
    meta = json.load(open(os.path.join(GOLD, "config1.json")))
    frames = e.synth.synth_frames(meta["seed"], meta["classes"], P, 0, meta["T"])
    f = tmp_path / "data" / "predictors" / "_" / "all.prd"
    e.formats.write_prd(str(f), "_", frames)
    monkeypatch.setenv("ECOZ2_VQ_OUT_ROOT", str(tmp_path))
    monkeypatch.setenv("ECOZ2_VQ_MAX_CODEBOOK_SIZE", str(meta["max_M"]))
    monkeypatch.setenv("ECOZ2_VQ_GPUS", str(ranks))
    seen = []
    e.vq_learn(None, P, meta["eps"], "_", [str(f)], callback=lambda *a: seen.append(a))
    assert [s[0] for s in seen] == [g["M"] for g in meta["levels"]]
    for s, g in zip(seen, meta["levels"]):
        assert (s[1].hex(), s[2].hex(), s[3].hex()) == (g["avg"], g["sigma"], g["inertia"])
        name = f"eps_0.05_M_{g['M']:04d}.cbook"
        got = open(tmp_path / "data" / "codebooks" / "_" / name, "rb").read()
        assert got == open(os.path.join(GOLD, "config1_" + name), "rb").read()


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus N` with no launcher around it: the parent starts the N ranks itself (fresh child
    processes, before any GPU call of its own), relays rank 0's JSON line and reports the collective.  Two ranks share
    the one GPU here, so the exchange is staged through gloo; on an N-GPU node the same command runs nccl = RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
                        "--frames-per-gpu", str(1 << 20)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    c = d["config"]["collective"]
    assert c["world_size"] == 2 and c["devices_per_rank"] == 1 and c["backend"].startswith("gloo")
    assert c["allreduce_calls"] >= 6 and c["bytes_per_call"] == 1024 * e.lib.e2vq_row_stride(P) * 8
    # a rank that fails makes the launcher fail
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "no-such-backend", "--steps", "3",
                        "--frames-per-gpu", "4096"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_bench_in_process_group_and_parity_section():
    """`bench.py --gpus 2 --in-process`: the two ranks are host threads driving one session each through the library's OWN
    in-process group (e2vq_group_*: what ecoz2_vq_learn runs for ECOZ2_VQ_GPUS=2) -- on this 1-GPU box both ranks share the
    device, so the exchange is the library's peer-to-peer kernel; on a node with a GPU per rank the same command times
    ncclAllReduce on the RCCL the library loads.  The line carries the exchange's device time per call and the untimed parity
    section: the timed level re-run on the plain FP64 sweep gives the same codebook, and the timed kernel's per-frame outputs
    equal the strict oracle's on a sample."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--in-process",
                        "--steps", "6", "--frames-per-gpu", str(1 << 18), "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    c = d["config"]["collective"]
    assert c["world_size"] == 2 and "in-process group" in c["exchange"] and "rank(s)" in c["library_says"]
    assert c["allreduce_calls_timed"] >= 6 and c["allreduce_us_per_call"] > 0
    assert c["allreduce_bytes_per_call"] == 1024 * e.lib.e2vq_row_stride(P) * 8
    par = d["config"]["parity"]
    assert par["ok"] and par["equals_plain_sweep"] is True and par["oracle_mismatches"] == 0 and par["oracle_sample_frames"] == 32768
    # one rank, process mode: the same section, with the cell counts checked against the symbols as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--frames-per-gpu", str(1 << 17), "--no-extras",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    par = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["config"]["parity"]
    assert par["ok"] and par["equals_plain_sweep"] is True and par["oracle_mismatches"] == 0 and par["cell_counts_match_symbols"] is True


def test_rccl_inside_the_library_single_rank_group(tmp_path):
    """ECOZ2_VQ_COLLECTIVE=rccl: libecoz2vq.so loads librccl.so itself (dlopen), builds a communicator with
    ncclCommInitAll over the in-process ranks' devices and all-reduces the int64 cell sums with ncclAllReduce on the
    session's stream -- here a group of ONE rank (all a 1-GPU box allows: RCCL wants one device per rank), through
    ecoz2_vq_learn, with the hook forced: the golden codebooks and callback scalars must come out unchanged.
    Runs in a process of its own, as the library runs under the reference's Rust host: this test process has PyTorch's
    bundled ROCm libraries mapped beside /opt/rocm's, and an RCCL initialised in that mix finds no device."""
    meta = json.load(open(os.path.join(GOLD, "config1.json")))
    frames = e.synth.synth_frames(meta["seed"], meta["classes"], P, 0, meta["T"])
    f = tmp_path / "data" / "predictors" / "_" / "all.prd"
    e.formats.write_prd(str(f), "_", frames)
    env = dict(os.environ, ECOZ2_VQ_OUT_ROOT=str(tmp_path), ECOZ2_VQ_MAX_CODEBOOK_SIZE=str(meta["max_M"]), ECOZ2_VQ_GPUS="1",
               ECOZ2_VQ_COLLECTIVE="rccl", NCCL_DEBUG="WARN")
    env.pop("ECOZ2_VQ_QUIET", None)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import ecoz2rs_amd as e; seen = []; "
            f"e.vq_learn(None, {P}, {meta['eps']!r}, '_', [{str(f)!r}], callback=lambda *a: seen.append(a)); "
            "print('SEEN', [(s[0], s[1].hex(), s[2].hex(), s[3].hex()) for s in seen])")

    def run(gpus):
        env["ECOZ2_VQ_GPUS"] = str(gpus)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout

    out = run(1)
    assert "collective: RCCL" in out and "ncclAllReduce(int64 sum)" in out
    calls = [ln for ln in out.splitlines() if "ncclAllReduce call(s)" in ln]
    # one MAX (two words) + one SUM at e2vq_prepare, then one SUM of the rows per pass
    assert calls and int(calls[0].split("made")[1].split()[0]) >= 2 + sum(g["passes"] for g in meta["levels"])
    seen = eval([ln for ln in out.splitlines() if ln.startswith("SEEN ")][0][5:])
    assert seen == [(g["M"], g["avg"], g["sigma"], g["inertia"]) for g in meta["levels"]]
    for g in meta["levels"]:
        name = f"eps_0.05_M_{g['M']:04d}.cbook"
        got = open(tmp_path / "data" / "codebooks" / "_" / name, "rb").read()
        assert got == open(os.path.join(GOLD, "config1_" + name), "rb").read()
    # two ranks on the one device: RCCL refuses that placement, the library says so and uses the peer-to-peer kernel
    if e.lib.e2vq_device_count() < 2:
        out = run(2)
        assert "RCCL needs one device per rank" in out and "peer-to-peer" in out
        got = open(tmp_path / "data" / "codebooks" / "_" / "eps_0.05_M_0016.cbook", "rb").read()
        assert got == open(os.path.join(GOLD, "config1_eps_0.05_M_0016.cbook"), "rb").read()


@pytest.mark.parametrize("collective", ["p2p", "rccl"])
def test_bad_shard_fails_every_rank_of_a_group(tmp_path, collective):
    """One rank of an in-process group finds NaN in ITS shard.  The bad-data flags are all-reduced at e2vq_prepare, every
    collective starts with a host rendezvous of the group, and a failed group aborts its RCCL communicators: the call must
    come back with the data error, not leave the healthy ranks blocked in a collective nobody else will join.  (p2p: three
    ranks sharing the GPU; rccl: the one-rank communicator a 1-GPU box allows, or one rank per device where there are more.)
    A subprocess with a timeout: a hang is the failure this guards against."""
    frames = e.synth.synth_frames(71, 4, P, 0, 30000)
    frames[29000, 7] = np.nan  # in the last rank's range
    f = tmp_path / "data" / "predictors" / "_" / "all.prd"
    e.formats.write_prd(str(f), "_", frames)
    ranks = 3 if collective == "p2p" else max(1, min(3, e.lib.e2vq_device_count()))
    env = dict(os.environ, ECOZ2_VQ_OUT_ROOT=str(tmp_path), ECOZ2_VQ_MAX_CODEBOOK_SIZE="64", ECOZ2_VQ_GPUS=str(ranks),
               ECOZ2_VQ_COLLECTIVE=collective, ECOZ2_VQ_QUIET="1")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import ecoz2rs_amd as e\n"
            f"try:\n    e.vq_learn(None, {P}, 0.05, '_', [{str(f)!r}])\n    print('NO ERROR')\n"
            "except Exception as ex:\n    print('ERROR:', ex)\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ERROR:" in r.stdout and "NaN or infinite" in r.stdout, r.stdout[-2000:]
    assert not list((tmp_path / "data").glob("codebooks/_/*.cbook"))


@pytest.mark.parametrize("chunk", [None, 4096])
@pytest.mark.parametrize("workers", [1, 2, 5])
def test_quantize_workers_give_identical_seq_files(tmp_path, monkeypatch, capfd, oracle, workers, chunk):
    """ecoz2_vq_quantize with ECOZ2_VQ_GPUS=N (frames are independent -- files dealt to N workers, no
    collective; the workers share the one GPU here): every .seq byte-identical to the oracle's symbols whatever N, totals
    summed in file order.  Ragged files, an empty file, more workers than some ranks have files."""
    M = 64
    frames = e.synth.synth_frames(808, 4, P, 0, 30000)
    refl = np.zeros((M, P + 1))
    for i in range(M):
        refl[i, 1:] = oracle.lpca_r(frames[i * 401], P)[2][1:]
    cb = tmp_path / "data" / "codebooks" / "_" / "cb.cbook"
    e.formats.write_cbook(str(cb), "_", refl)
    cuts = [0, 1, 500, 500, 4097, 9000, 17000, 17064, 30000]  # (one empty file)
    files = []
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        f = tmp_path / "data" / "predictors" / f"c{i % 3}" / f"{i:05d}.prd"
        e.formats.write_prd(str(f), f"c{i % 3}", frames[a:b])
        files.append(str(f))
    monkeypatch.setenv("ECOZ2_VQ_OUT_ROOT", str(tmp_path))
    monkeypatch.setenv("ECOZ2_VQ_GPUS", str(workers))
    if chunk:  # small units: short files are batched into one sweep, the long ones split over the workers
        monkeypatch.setenv("ECOZ2_VQ_QUANTIZE_CHUNK", str(chunk))
    capfd.readouterr()
    e.vq_quantize(str(cb), files, True)
    out = capfd.readouterr().out
    sym_o, dmin_o = oracle.quantize(oracle.reflections_to_cq(refl), frames)
    total = 0.0
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        cls, m, sym = e.formats.read_seq(str(tmp_path / "data" / "sequences" / f"M{M}" / f"c{i % 3}" / f"{i:05d}.seq"))
        assert (cls, m) == (f"c{i % 3}", M) and np.array_equal(sym, sym_o[a:b])
        acc = 0.0
        for d in dmin_o[a:b]:
            acc += d - 1.0
        total += acc
        assert f"{files[i]}: 'c{i % 3}' T={b - a} avg distortion={(acc / (b - a) if b > a else 0.0):g} ->" in out
    assert f"total: 8 predictor file(s), 30000 vectors, M={M}, avg distortion={total / 30000:g}" in out
    # non-finite input is refused (the file entry points check it; the sweep's argmin assumes finite data)
    bad = frames[:10].copy()
    bad[3, 5] = np.nan
    fb = tmp_path / "bad.prd"
    e.formats.write_prd(str(fb), "x", bad)
    with pytest.raises(e.Ecoz2Error, match="NaN or infinite"):
        e.vq_quantize(str(cb), files[:2] + [str(fb)])


def test_restored_level_repeats_the_uninterrupted_ladder(oracle):
    """bench.py times the real M = 1024 level by restoring the converged M = 512 codebook and its DD
    (e2vq_set_codebook + e2vq_set_prev_distortion) and running the level again through e2vq_learn: the repetition must be
    the level the uninterrupted ladder ran -- same passes, DD, statistics, codebook -- every time (here 128 -> 256)."""
    frames = e.synth.synth_frames(909, 6, P, 0, 40000)
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    with e.VqSession(P) as s:
        s.set_frames(frames)
        s.prepare()
        s.init_codebook()
        levels = s.learn(0.05, 128)
        cb128, dd128 = s.get_codebook(), levels[-1].DD
        assert s.prev_distortion() == dd128
        ref = s.learn(0.05, 256)[0]
        cb256 = s.get_codebook()
        for _ in range(3):
            s.set_codebook(cb128)
            s.set_prev_distortion(dd128)
            again = s.learn(0.05, 256)[0]
            assert again == ref and np.array_equal(s.get_codebook().view(np.uint64), cb256.view(np.uint64))
        # without the DD the stopping rule sees another ratio on pass 1 (it may or may not change the pass count,
        # but DDprv must be what was set)
        s.set_codebook(cb128)
        s.set_prev_distortion(1.0)
        assert s.prev_distortion() == 1.0
    rc, levels_o, _ = oracle.learn(frames, 0.05, 256)
    assert (ref.passes, ref.DD) == (levels_o[-1]["passes"], levels_o[-1]["DD"])
    assert np.array_equal(cb256.view(np.uint64), levels_o[-1]["reflections"].view(np.uint64))
