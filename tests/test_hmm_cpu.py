"""HMM consumers, CPU side: the oracle (oracle/hmm_oracle.c) against an independent log-domain
restatement and the Baum-Welch invariants, and the product's host pieces that need no GPU (generator, initial
models, .hmm files, `hmm show`, symbol exports).  The GPU parity tests are in test_gpu_hmm.py."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import ecoz2rs_amd as e
from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    return oracle_lib.load_hmm()


def _seqs(rng, M, n, lo=5, hi=60):
    return [rng.integers(0, M, rng.integers(lo, hi)).astype(np.uint16) for _ in range(n)]


def _naive_log_forward(pi, A, B, s):
    with np.errstate(divide="ignore"):
        la = np.log(pi) + np.log(B[:, s[0]])
        for t in range(1, len(s)):
            la = np.logaddexp.reduce(la[:, None] + np.log(A), axis=0) + np.log(B[:, s[t]])
    return np.logaddexp.reduce(la)


@pytest.mark.parametrize("typ", [0, 1, 2, 3])
def test_oracle_forward_and_baum_welch_invariants(H, typ):
    H.seed(7)
    N, M = 5, 32
    rng = np.random.default_rng(typ)
    pi, A, B = H.init(N, M, typ)
    assert abs(pi.sum() - 1) < 1e-12 and np.allclose(A.sum(1), 1) and np.allclose(B.sum(1), 1)
    if typ >= 2:  # cascade: upper band of width 2 / 3, start in state 0
        assert pi[0] == 1 and np.all(np.tril(A, -1) == 0) and np.all(np.triu(A, typ) == 0)
    seqs = _seqs(rng, M, 30)
    for s in seqs[:6]:
        st, m, ex = H.forward(pi, A, B, s)
        ref = _naive_log_forward(pi, A, B, s)
        assert st == 0 and 0.5 <= m < 1 and abs(H.log_prob(m, ex) - ref) < 1e-9 * abs(ref)
    p2, A2, B2, hist = H.learn(pi, A, B, seqs, 1e-5, 0.3, -1)
    assert len(hist) >= 2 and all(b >= a - 1e-6 for a, b in zip(hist, hist[1:]))  # EM never decreases sum ln P
    assert hist[-1] - hist[-2] <= 0.3  # the stopping rule
    assert abs(p2.sum() - 1) < 1e-9 and np.allclose(A2.sum(1), 1, atol=1e-9) and np.allclose(B2.sum(1), 1, atol=1e-9)
    assert B2.min() >= 1e-5 * 0.99  # epsilon restriction on B
    # the accumulators are exact integer sums: any order / partition of the sequences gives the same words
    acc_all, _ = H.accumulate(pi, A, B, seqs)
    acc_a, _ = H.accumulate(pi, A, B, seqs[17:])
    acc_b, _ = H.accumulate(pi, A, B, seqs[:17][::-1])
    assert np.array_equal(acc_all, acc_a + acc_b)
    # expected counts: sum_i gamma_0(i) = #sequences, sum over states of BD = #symbols (up to rounding of each term)
    sh = 2.0 ** -(29 + 31)
    val = lambda k: (int(acc_all[2 * k]) * 2 ** 31 + int(acc_all[2 * k + 1])) * sh
    assert abs(sum(val(i) for i in range(N)) - len(seqs)) < 1e-9
    off_bd = N + N * N + N + N * M
    assert abs(sum(val(off_bd + j) for j in range(N)) - sum(len(s) for s in seqs)) < 1e-8
    assert acc_all[-2] == len(seqs) and acc_all[-1] == 0


def test_oracle_max_iterations_and_impossible_sequences(H):
    H.seed(3)
    pi, A, B = H.init(4, 8, 3)
    rng = np.random.default_rng(0)
    seqs = _seqs(rng, 8, 10)
    _p, _A, _B, hist = H.learn(pi, A, B, seqs, 1e-5, 0.0, 4)
    assert len(hist) == 4
    _p, _A, _B, hist0 = H.learn(pi, A, B, seqs, 1e-5, 0.0, 0)
    assert hist0 == [] and np.array_equal(_B, B)
    B0 = B.copy()
    B0[:, 5] = 0.0  # no state emits symbol 5
    bad = np.array([1, 5, 2], dtype=np.uint16)
    st, m, ex = H.forward(pi, A, B0, bad)
    assert st == 1 and m == 0.0 and H.log_prob(m, ex) == -np.inf
    acc, res = H.accumulate(pi, A, B0, [bad, seqs[0][seqs[0] != 5]])
    assert [r[0] for r in res] == [1, 0] and acc[-2] == 1 and acc[-1] == 1  # used, skipped
    assert H.forward(pi, A, B, np.array([9], dtype=np.uint16))[0] == 2  # symbol outside the alphabet


@pytest.mark.parametrize("typ", [0, 1, 2, 3])
def test_product_generator_and_files_equal_the_oracle(H, typ, tmp_path):
    """ecoz2_set_random_seed + the initial models of `hmm learn -t`, and the .hmm layout: host code of the product
    (no GPU needed) against the oracle, bit for bit"""
    assert H.seed(12345) == 12345 and e.hmm.set_random_seed(12345) == 12345
    N, M = 6, 40
    po, Ao, Bo = H.init(N, M, typ)
    pp, Ap, Bp = e.hmm.init_model(N, M, typ)
    for a, b in ((po, pp), (Ao, Ap), (Bo, Bp)):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    fo, fp = tmp_path / "o.hmm", tmp_path / "p.hmm"
    H.save(fo, "Some class", po, Ao, Bo)
    e.hmm.save_model(fp, "Some class", pp, Ap, Bp)
    raw = open(fp, "rb").read()
    assert raw == open(fo, "rb").read()
    assert raw[:5] == b"<hmm>" and raw[16:26] == b"Some class" and raw[112:120] == bytes([N, 0, 0, 0, M, 0, 0, 0])
    assert len(raw) == 120 + 8 * (N + N * N + N * M)
    cls, p2, A2, B2 = e.hmm.load_model(fo)
    assert cls == "Some class" and np.array_equal(B2, Bo) and np.array_equal(A2, Ao) and np.array_equal(p2, po)
    assert H.load(fp)[0] == "Some class"
    (tmp_path / "bad.hmm").write_bytes(b"<sequence>" + b"\0" * 300)
    with pytest.raises(e.Ecoz2Error, match="Not an HMM model"):
        e.hmm.load_model(tmp_path / "bad.hmm")
    with pytest.raises(e.Ecoz2Error):
        e.hmm.init_model(513, 8, 1)  # N <= 512 (up to 64 a lane per state, beyond a thread of a workgroup)
    assert e.hmm.set_random_seed(-1) > 1_600_000_000  # negative: time based (src/hmm/mod.rs:73-76)


def test_hmm_show_cli(H, tmp_path):
    H.seed(1)
    pi, A, B = H.init(3, 4, 3)
    f = tmp_path / "m.hmm"
    H.save(f, "Bd", pi, A, B)
    exe = os.path.join(ROOT, "ecoz2rs_amd", "csrc", "ecoz2")
    r = subprocess.run([exe, "hmm", "show", "--hmm", str(f)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    lines = r.stdout.splitlines()
    assert lines[0] == f"hmm_show: hmm_filename={f} format=%Lg "  # src/ecoz2_lib/mod.rs:482-486
    assert "className='Bd', N=3, M=4" in lines[2]
    assert lines[3] == "pi = 1 0 0 "
    assert lines[5] == " [0]: " + "".join("%g " % v for v in A[0])
    assert lines[-1] == " [2]: " + "".join("%g " % v for v in B[2])
    r = subprocess.run([exe, "hmm", "show", "--hmm", str(f), "--format", "%.3f,"], capture_output=True, text=True, timeout=60)
    assert " [1]: " + "".join("%.3f," % v for v in A[1]) in r.stdout.splitlines()


def test_classify_header_symbols_are_exported_and_need_a_device():
    header = open(os.path.join(ROOT, "include", "ecoz2_classify.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b((?:ecoz2|e2vq)_[a-z0-9_]+)\s*\(", header)) - {"ecoz2_hmm_learn_callback_t"}
    assert {"ecoz2_hmm_learn", "ecoz2_hmm_classify", "ecoz2_hmm_classify_predictors", "ecoz2_hmm_show",
            "ecoz2_set_random_seed", "ecoz2_nb_learn", "ecoz2_mm_classify"} <= declared
    lib = C.CDLL(e.lib_path)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    if e.lib.e2vq_device_count() > 0:
        pytest.skip("a HIP device is present")
    # no CPU fallback: scoring / training fail loudly without a device
    pi, A, B = e.hmm.init_model(3, 4, 1)
    with pytest.raises(e.Ecoz2Error, match="no HIP device"):
        e.hmm.score([(pi, A, B)], [np.array([0, 1], dtype=np.uint16)])
    with pytest.raises(e.Ecoz2Error, match="no HIP device"):
        e.hmm.train(pi, A, B, [np.array([0, 1], dtype=np.uint16)])
