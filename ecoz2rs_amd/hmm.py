"""Host-side mirror of the reference's HMM wrappers.

``hmm_learn`` / ``hmm_classify_sequences`` / ``hmm_classify_predictors`` / ``hmm_show`` / ``set_random_seed`` take the
arguments of the Rust functions of the same name (/root/reference/src/ecoz2_lib/mod.rs:188-190, 385-494) and call the
same C symbols; the array-level functions drive the same HIP kernels without files (tests, bench).
"""
import ctypes as C

import numpy as np

from ._lib import c_char_pp, check, lib

HMM_LEARN_CALLBACK = C.CFUNCTYPE(None, C.c_char_p, C.c_double)
_dpp = C.POINTER(C.c_void_p)


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("ecoz2_set_random_seed", C.c_ulong, C.c_long)
_sig("ecoz2_hmm_learn", C.c_int, C.c_int, C.c_int, c_char_pp, C.c_uint, C.c_double, C.c_double, C.c_int, C.c_int,
     HMM_LEARN_CALLBACK)
_sig("ecoz2_hmm_classify", C.c_int, c_char_pp, C.c_uint, c_char_pp, C.c_uint, C.c_int, C.c_char_p)
_sig("ecoz2_hmm_classify_predictors", C.c_int, c_char_pp, C.c_uint, c_char_pp, C.c_int, c_char_pp, C.c_int, C.c_int,
     C.c_char_p)
_sig("ecoz2_hmm_show", C.c_int, C.c_char_p, C.c_char_p)
_sig("e2vq_hmm_init", C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("e2vq_hmm_save", C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("e2vq_hmm_info", C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("e2vq_hmm_load", C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("e2vq_hmm_score", C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _dpp, _dpp, _dpp, C.c_void_p, C.c_void_p,
     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("e2vq_hmm_acc_words", C.c_int64, C.c_int, C.c_int)
_sig("e2vq_hmm_estep", C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("e2vq_hmm_train", C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
     C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int))


def _strs(items):
    arr = (C.c_char_p * len(items))(*[str(p).encode() for p in items])
    return C.cast(arr, c_char_pp), arr


def set_random_seed(seed):
    """ecoz2_lib::set_random_seed (src/ecoz2_lib/mod.rs:188-190)"""
    return lib.ecoz2_set_random_seed(int(seed))


def hmm_learn(n, model_type, sequence_filenames, hmm_epsilon, val_auto, max_iterations, use_par=True, callback=None):
    """ecoz2_lib::hmm_learn (src/ecoz2_lib/mod.rs:385-419); callback(var: str, val: float) per E-step"""
    files, _k = _strs(sequence_filenames)
    cb = HMM_LEARN_CALLBACK((lambda v, x: callback(v.decode(), x)) if callback else (lambda _v, _x: None))
    check(lib.ecoz2_hmm_learn(int(n), int(model_type), files, len(sequence_filenames), float(hmm_epsilon),
                              float(val_auto), int(max_iterations), int(bool(use_par)), cb))


def hmm_classify_sequences(model_filenames, sequence_filenames, show_ranked=False, classification_filename=None):
    """ecoz2_lib::hmm_classify_sequences (src/ecoz2_lib/mod.rs:421-447)"""
    m, _k1 = _strs(model_filenames)
    s, _k2 = _strs(sequence_filenames)
    check(lib.ecoz2_hmm_classify(m, len(model_filenames), s, len(sequence_filenames), int(show_ranked),
                                 str(classification_filename).encode() if classification_filename else None))


def hmm_classify_predictors(model_filenames, cb_filenames, prd_filenames, show_ranked=False,
                            classification_filename=None):
    """ecoz2_lib::hmm_classify_predictors (src/ecoz2_lib/mod.rs:449-479)"""
    m, _k1 = _strs(model_filenames)
    c, _k2 = _strs(cb_filenames)
    p, _k3 = _strs(prd_filenames)
    check(lib.ecoz2_hmm_classify_predictors(m, len(model_filenames), c, len(cb_filenames), p, len(prd_filenames),
                                            int(show_ranked),
                                            str(classification_filename).encode() if classification_filename else None))


def hmm_show(hmm_filename, format="%Lg "):
    """ecoz2_lib::hmm_show (src/ecoz2_lib/mod.rs:481-494)"""
    check(lib.ecoz2_hmm_show(str(hmm_filename).encode(), format.encode()))


# ---- array level ----------------------------------------------------------------------------------------------
def init_model(N, M, model_type):
    pi, A, B = np.zeros(N), np.zeros((N, N)), np.zeros((N, M))
    check(lib.e2vq_hmm_init(N, M, model_type, pi.ctypes.data, A.ctypes.data, B.ctypes.data))
    return pi, A, B


def save_model(path, class_name, pi, A, B):
    pi, A, B = (np.ascontiguousarray(x, dtype=np.float64) for x in (pi, A, B))
    check(lib.e2vq_hmm_save(str(path).encode(), class_name.encode(), len(pi), B.shape[1], pi.ctypes.data, A.ctypes.data,
                            B.ctypes.data))


def load_model(path):
    cls, N, M = C.create_string_buffer(96), C.c_int(), C.c_int()
    check(lib.e2vq_hmm_info(str(path).encode(), cls, C.byref(N), C.byref(M)))
    pi, A, B = np.zeros(N.value), np.zeros((N.value, N.value)), np.zeros((N.value, M.value))
    check(lib.e2vq_hmm_load(str(path).encode(), pi.ctypes.data, A.ctypes.data, B.ctypes.data))
    return cls.value.decode(), pi, A, B


def _pack(seqs):
    arrs = [np.ascontiguousarray(s, dtype=np.uint16) for s in seqs]
    offs = np.zeros(len(arrs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(a) for a in arrs])
    sym = np.concatenate(arrs) if arrs and offs[-1] else np.zeros(0, dtype=np.uint16)
    return np.ascontiguousarray(sym), offs


def score(models, seqs, device=0):
    """models: list of (pi, A, B) sharing M; seqs: list of uint16 arrays -> dict of (S, K) arrays mant/exp2/status/log_prob"""
    K, S = len(models), len(seqs)
    ms = [tuple(np.ascontiguousarray(x, dtype=np.float64) for x in m) for m in models]
    Ns = (C.c_int * K)(*[len(m[0]) for m in ms])
    ptr = lambda i: (C.c_void_p * K)(*[m[i].ctypes.data for m in ms])
    sym, offs = _pack(seqs)
    mant, ex = np.zeros((S, K)), np.zeros((S, K), dtype=np.int64)
    st, lp = np.zeros((S, K), dtype=np.int32), np.zeros((S, K))
    check(lib.e2vq_hmm_score(device, K, Ns, ms[0][2].shape[1], ptr(0), ptr(1), ptr(2), sym.ctypes.data, offs.ctypes.data,
                             S, mant.ctypes.data, ex.ctypes.data, st.ctypes.data, lp.ctypes.data))
    return dict(mant=mant, exp2=ex, status=st, log_prob=lp)


def estep(pi, A, B, seqs, device=0):
    """one Baum-Welch E-step: (acc int64 words, mant, exp2, status)"""
    pi, A, B = (np.ascontiguousarray(x, dtype=np.float64) for x in (pi, A, B))
    N, M, S = len(pi), B.shape[1], len(seqs)
    sym, offs = _pack(seqs)
    acc = np.zeros(lib.e2vq_hmm_acc_words(N, M), dtype=np.int64)
    mant, ex, st = np.zeros(S), np.zeros(S, dtype=np.int64), np.zeros(S, dtype=np.int32)
    check(lib.e2vq_hmm_estep(device, N, M, pi.ctypes.data, A.ctypes.data, B.ctypes.data, sym.ctypes.data, offs.ctypes.data,
                             S, acc.ctypes.data, mant.ctypes.data, ex.ctypes.data, st.ctypes.data))
    return acc, mant, ex, st


def train(pi, A, B, seqs, epsilon=1e-5, val_auto=0.3, max_iterations=-1, device=0):
    """Baum-Welch on arrays: -> (pi, A, B, [sum_log_prob per E-step])"""
    pi, A, B = (np.array(x, dtype=np.float64, copy=True) for x in (pi, A, B))
    N, M, S = len(pi), B.shape[1], len(seqs)
    sym, offs = _pack(seqs)
    hist, n = np.zeros(4096), C.c_int()
    check(lib.e2vq_hmm_train(device, N, M, pi.ctypes.data, A.ctypes.data, B.ctypes.data, sym.ctypes.data, offs.ctypes.data,
                             S, float(epsilon), float(val_auto), int(max_iterations), hist.ctypes.data, len(hist),
                             C.byref(n)))
    return pi, A, B, list(hist[:n.value])
