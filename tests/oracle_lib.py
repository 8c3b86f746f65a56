"""ctypes binding of the CPU oracle (oracle/_build/libvqoracle.so).  TESTS ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MAX_P = 200


class Stats(C.Structure):
    _fields_ = [
        ("maxabs", C.c_double),
        ("sum_hi", C.c_int64 * (MAX_P + 1)),
        ("sum_lo", C.c_int64 * (MAX_P + 1)),
        ("q_hi", C.c_int64),
        ("q_lo", C.c_int64),
    ]


class LevelStats(C.Structure):
    _fields_ = [
        ("DD", C.c_double),
        ("avg", C.c_double),
        ("sigma", C.c_double),
        ("inertia", C.c_double),
        ("empty_cells", C.c_int64),
        ("failed_cells", C.c_int64),
    ]


LEARN_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double)
LEVEL_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(LevelStats))


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        L.e2o_distortion.restype = C.c_double
        L.e2o_unfix.restype = C.c_double
        L.e2o_unfix.argtypes = [C.c_int64, C.c_int64, C.c_int]
        L.e2o_fix.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.e2o_shift_frames.argtypes = [C.c_double]
        L.e2o_shift_frames_sq.argtypes = [C.c_double]
        L.e2o_dist_exponent.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_double]
        L.e2o_quantize.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.e2o_pass.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p]
        L.e2o_data_stats.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.POINTER(Stats)]
        L.e2o_rows_stats.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double,
                                     C.POINTER(LevelStats)]
        L.e2o_update.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(LevelStats)]
        L.e2o_grow.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.e2o_reflections_to_cq.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.e2o_lpca.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_double)]
        L.e2o_lpca_r.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        L.e2o_ref2raas.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.e2o_learn.argtypes = [C.c_int, C.c_double, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                C.c_char_p, C.c_void_p, LEARN_CB, C.c_void_p, LEVEL_HOOK]
        L.e2o_prd_save.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]
        L.e2o_cbook_save.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.e2o_seq_save.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]
        L.e2o_time_pass.restype = C.c_double
        L.e2o_time_pass.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int)]

    # ---- small pieces
    def lpca(self, x, P):
        x = np.ascontiguousarray(x, dtype=np.float64)
        r, rc, a = (np.zeros(P + 1) for _ in range(3))
        pe = C.c_double()
        st = self.L.e2o_lpca(_dp(x), len(x), P, _dp(r), _dp(rc), _dp(a), C.byref(pe))
        return st, pe.value, r, rc, a

    def lpca_r(self, r, P):
        r = np.ascontiguousarray(r, dtype=np.float64)
        rc, a = np.zeros(P + 1), np.zeros(P + 1)
        pe = C.c_double()
        st = self.L.e2o_lpca_r(P, _dp(r), _dp(rc), _dp(a), C.byref(pe))
        return st, pe.value, rc, a

    def ref2raas(self, rc):
        rc = np.ascontiguousarray(rc, dtype=np.float64)
        raa = np.zeros_like(rc)
        self.L.e2o_ref2raas(len(rc) - 1, _dp(rc), _dp(raa))
        return raa

    def reflections_to_cq(self, refl):
        refl = np.ascontiguousarray(refl, dtype=np.float64)
        cq = np.zeros_like(refl)
        self.L.e2o_reflections_to_cq(refl.shape[1] - 1, refl.shape[0], _dp(refl), _dp(cq))
        return cq

    def row_stride(self, P):
        return self.L.e2o_row_stride(P)

    def data_stats(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        st = Stats()
        rc = self.L.e2o_data_stats(frames.shape[1] - 1, _dp(frames), frames.shape[0], C.byref(st))
        return rc, st

    def shifts(self, maxabs):
        return self.L.e2o_shift_frames(maxabs), self.L.e2o_shift_frames_sq(maxabs)

    def dist_exponent(self, cq, maxabs):
        cq = np.ascontiguousarray(cq, dtype=np.float64)
        return self.L.e2o_dist_exponent(cq.shape[1] - 1, _dp(cq), cq.shape[0], maxabs)

    def unfix(self, hi, lo, sh):
        return self.L.e2o_unfix(int(hi), int(lo), int(sh))

    # ---- passes
    def quantize(self, cq, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        cq = np.ascontiguousarray(cq, dtype=np.float64)
        T = frames.shape[0]
        sym, dmin = np.zeros(T, dtype=np.uint16), np.zeros(T)
        self.L.e2o_quantize(frames.shape[1] - 1, _dp(cq), cq.shape[0], _dp(frames), T, _dp(sym), _dp(dmin))
        return sym, dmin

    def run_pass(self, cq, frames, sh_r, Ed):
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        cq = np.ascontiguousarray(cq, dtype=np.float64)
        P, T, M = frames.shape[1] - 1, frames.shape[0], cq.shape[0]
        sym, dmin = np.zeros(T, dtype=np.uint16), np.zeros(T)
        rows = np.zeros((M, self.row_stride(P)), dtype=np.int64)
        self.L.e2o_pass(P, _dp(cq), M, _dp(frames), T, sh_r, Ed, _dp(sym), _dp(dmin), _dp(rows))
        return sym, dmin, rows

    def rows_stats(self, rows, P, T, sh_r, Ed, Q):
        out = LevelStats()
        self.L.e2o_rows_stats(P, rows.shape[0], _dp(rows), T, sh_r, Ed, Q, C.byref(out))
        return out

    def update(self, rows, P, sh_r, refl):
        refl = np.array(refl, dtype=np.float64, copy=True)
        out = LevelStats()
        self.L.e2o_update(P, rows.shape[0], _dp(rows), sh_r, _dp(refl), C.byref(out))
        return refl, out.failed_cells

    def grow(self, refl):
        refl = np.ascontiguousarray(refl, dtype=np.float64)
        out = np.zeros((2 * refl.shape[0], refl.shape[1]))
        self.L.e2o_grow(refl.shape[1] - 1, refl.shape[0], _dp(refl), _dp(out))
        return out

    def learn(self, frames, eps, max_M, class_name="_", base=None, out_root=None):
        """Returns list of dicts per level: M, passes, reflections, DD, avg, sigma, inertia, empty; and callbacks."""
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        P = frames.shape[1] - 1
        levels, cbs = [], []

        def hook(_u, M, passes, refl, st):
            r = np.ctypeslib.as_array(refl, shape=(M, P + 1)).copy()
            s = st.contents
            levels.append(dict(M=M, passes=passes, reflections=r, DD=s.DD, avg=s.avg, sigma=s.sigma,
                               inertia=s.inertia, empty=s.empty_cells))

        def cb(_t, M, avg, sigma, inertia):
            cbs.append((M, avg, sigma, inertia))

        base_a = np.ascontiguousarray(base, dtype=np.float64) if base is not None else None
        rc = self.L.e2o_learn(P, eps, class_name.encode(), _dp(frames), frames.shape[0],
                              _dp(base_a) if base_a is not None else None, base_a.shape[0] if base_a is not None else 0,
                              max_M, out_root.encode() if out_root else None, None, LEARN_CB(cb), None,
                              LEVEL_HOOK(hook))
        return rc, levels, cbs

    def set_threads(self, n):
        self.L.e2o_set_threads(int(n))

    def time_pass(self, cq, frames, reps=1):
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        cq = np.ascontiguousarray(cq, dtype=np.float64)
        nt = C.c_int()
        secs = self.L.e2o_time_pass(frames.shape[1] - 1, _dp(cq), cq.shape[0], _dp(frames), frames.shape[0], reps,
                                    C.byref(nt))
        return secs, nt.value


def build(target="all"):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, target], check=True)


def load(variant="libvqoracle.so"):
    path = os.path.join(ORACLE_DIR, "_build", variant)
    if not os.path.exists(path):
        build()
    return Oracle(path)


# ---------------------------------------------------------------------------------------------------------------
# HMM oracle (oracle/hmm_oracle.c, in the same shared object).  TESTS ONLY.
# ---------------------------------------------------------------------------------------------------------------
HMM_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_double)


class HmmOracle:
    ACC_SHIFT = 29

    def __init__(self, path=None):
        path = path or os.path.join(ORACLE_DIR, "_build", "libvqoracle.so")
        if not os.path.exists(path):
            build()
        L = self.L = C.CDLL(path)
        L.e2h_set_random_seed.restype = C.c_uint64
        L.e2h_set_random_seed.argtypes = [C.c_int64]
        L.e2h_uniform.restype = C.c_double
        L.e2h_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.e2h_forward.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
        L.e2h_log_prob.restype = C.c_double
        L.e2h_log_prob.argtypes = [C.c_double, C.c_int64]
        L.e2h_acc_words.restype = C.c_int64
        L.e2h_acc_words.argtypes = [C.c_int, C.c_int]
        L.e2h_accumulate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.e2h_reestimate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.e2h_learn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, HMM_CB]
        L.e2h_save.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.e2h_load_info.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.e2h_load.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def seed(self, s):
        return self.L.e2h_set_random_seed(int(s))

    def init(self, N, M, type_):
        pi, A, B = np.zeros(N), np.zeros((N, N)), np.zeros((N, M))
        assert self.L.e2h_init(N, M, type_, _dp(pi), _dp(A), _dp(B)) == 0
        return pi, A, B

    def forward(self, pi, A, B, sym, want_alpha=False):
        """-> (status, mant, exp2[, alpha_hat, c])"""
        sym = np.ascontiguousarray(sym, dtype=np.uint16)
        N, M, T = len(pi), B.shape[1], len(sym)
        mant, e2 = C.c_double(), C.c_int64()
        al = np.zeros((T, N)) if want_alpha else None
        c = np.zeros(T) if want_alpha else None
        st = self.L.e2h_forward(N, M, _dp(pi), _dp(A), _dp(B), _dp(sym), T, C.byref(mant), C.byref(e2),
                                _dp(al) if want_alpha else None, _dp(c) if want_alpha else None)
        return (st, mant.value, e2.value, al, c) if want_alpha else (st, mant.value, e2.value)

    def log_prob(self, mant, e2):
        return self.L.e2h_log_prob(mant, e2)

    def acc_words(self, N, M):
        return self.L.e2h_acc_words(N, M)

    def accumulate(self, pi, A, B, seqs):
        """E-step over the sequences: -> (acc int64 words, [(status, mant, exp2)...])"""
        N, M = len(pi), B.shape[1]
        acc = np.zeros(self.acc_words(N, M), dtype=np.int64)
        res = []
        for s in seqs:
            s = np.ascontiguousarray(s, dtype=np.uint16)
            mant, e2 = C.c_double(), C.c_int64()
            st = self.L.e2h_accumulate(N, M, _dp(pi), _dp(A), _dp(B), _dp(s), len(s), _dp(acc), C.byref(mant), C.byref(e2))
            res.append((st, mant.value, e2.value))
        return acc, res

    def reestimate(self, acc, eps, pi, A, B):
        pi, A, B = pi.copy(), A.copy(), B.copy()
        self.L.e2h_reestimate(len(pi), B.shape[1], _dp(acc), eps, _dp(pi), _dp(A), _dp(B))
        return pi, A, B

    def learn(self, pi, A, B, seqs, eps=1e-5, val_auto=0.3, max_iterations=-1):
        """-> (pi, A, B, [sum_log_prob per E-step])"""
        pi, A, B = pi.copy(), A.copy(), B.copy()
        N, M, R = len(pi), B.shape[1], len(seqs)
        arrs = [np.ascontiguousarray(s, dtype=np.uint16) for s in seqs]
        ptrs = (C.c_void_p * R)(*[a.ctypes.data for a in arrs])
        lens = np.array([len(a) for a in arrs], dtype=np.int64)
        hist = np.zeros(4096)
        n = self.L.e2h_learn(N, M, ptrs, _dp(lens), R, eps, val_auto, max_iterations, _dp(pi), _dp(A), _dp(B),
                             _dp(hist), len(hist), HMM_CB(lambda _v, _x: None))
        assert n >= 0
        return pi, A, B, list(hist[:n])

    def save(self, path, class_name, pi, A, B):
        assert self.L.e2h_save(str(path).encode(), class_name.encode(), len(pi), B.shape[1], _dp(pi), _dp(A), _dp(B)) == 0

    def load(self, path):
        cls, N, M = C.create_string_buffer(96), C.c_int(), C.c_int()
        assert self.L.e2h_load_info(str(path).encode(), cls, C.byref(N), C.byref(M)) == 0
        pi, A, B = np.zeros(N.value), np.zeros((N.value, N.value)), np.zeros((N.value, M.value))
        assert self.L.e2h_load(str(path).encode(), _dp(pi), _dp(A), _dp(B)) == 0
        return cls.value.decode(), pi, A, B


def load_hmm():
    return HmmOracle()


def rows_match(rows, rows_o, P):
    """GPU cell sums vs the oracle's: the limb sums and the count of every cell word for word; the four distortion
    elements (sum of e and of e^2 as limb pairs) by their COLUMN TOTALS -- the level statistics are all that ever reads
    them, and the prefiltered sweep adds each wave's distortion sums to the distortion columns of one row instead of four
    atomics per frame.  The oracle keeps them per cell; both give the same totals, exactly (integers)."""
    import numpy as np
    a, b = np.asarray(rows), np.asarray(rows_o)
    n = 2 * (P + 1) + 1
    return (a.shape == b.shape and np.array_equal(a[:, :n], b[:, :n])
            and np.array_equal(a[:, n:n + 4].sum(axis=0), b[:, n:n + 4].sum(axis=0)))
