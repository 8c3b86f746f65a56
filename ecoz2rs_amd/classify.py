"""Host-side mirror of the reference's `.seq` consumers: nb / mm / c12n.

Function names follow the Rust modules they mirror: ``nbayes::learn`` / ``classify`` (src/nb/nbayes.rs:63-153),
``markov::learn`` / ``classify`` (src/mm/markov.rs:59-167); the work happens in libecoz2vq.so (csrc/seq_models.cpp).
"""
import ctypes as C

from ._lib import c_char_pp, check, lib


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("e2vq_seq_info", C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int64))
_sig("e2vq_seq_read", C.c_int, C.c_char_p, C.c_void_p, C.c_int64)
_sig("ecoz2_nb_learn", C.c_int, C.c_int, c_char_pp, C.c_int, C.c_char_p, C.c_int)
_sig("ecoz2_nb_classify", C.c_int, c_char_pp, C.c_int, c_char_pp, C.c_int, C.c_int, C.c_int)
_sig("ecoz2_nb_show", C.c_int, C.c_char_p)
_sig("e2vq_nb_log_prob", C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_double))
_sig("ecoz2_mm_learn", C.c_int, C.c_int, c_char_pp, C.c_int, C.c_char_p, C.c_int)
_sig("ecoz2_mm_classify", C.c_int, c_char_pp, C.c_int, c_char_pp, C.c_int, C.c_int, C.c_int)
_sig("ecoz2_mm_show", C.c_int, C.c_char_p)
_sig("e2vq_mm_log_prob", C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_float))
_sig("e2vq_c12n_run", C.c_int, c_char_pp, C.c_int, C.POINTER(C.c_int), c_char_pp, c_char_pp, C.POINTER(C.c_double),
     C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int))


def _strs(items):
    arr = (C.c_char_p * len(items))(*[str(p).encode() for p in items])
    return C.cast(arr, c_char_pp), arr


def _learn(fn, codebook_size, seq_filenames):
    files, _k = _strs(seq_filenames)
    out = C.create_string_buffer(4096)
    check(fn(int(codebook_size), files, len(seq_filenames), out, 4096))
    return out.value.decode()


def nb_learn(codebook_size, seq_filenames):
    """nbayes::learn + main_nbayes_learn: returns the path of the saved `.nb` model"""
    return _learn(lib.ecoz2_nb_learn, codebook_size, seq_filenames)


def mm_learn(codebook_size, seq_filenames):
    """markov::learn + main_mm_learn: returns the path of the saved `.mm` model"""
    return _learn(lib.ecoz2_mm_learn, codebook_size, seq_filenames)


def _classify(fn, model_filenames, seq_filenames, show_ranked, codebook_size):
    m, _k1 = _strs(model_filenames)
    s, _k2 = _strs(seq_filenames)
    check(fn(m, len(model_filenames), s, len(seq_filenames), int(show_ranked), int(codebook_size)))


def nb_classify(nb_filenames, seq_filenames, show_ranked, codebook_size):
    _classify(lib.ecoz2_nb_classify, nb_filenames, seq_filenames, show_ranked, codebook_size)


def mm_classify(mm_filenames, seq_filenames, show_ranked, codebook_size):
    _classify(lib.ecoz2_mm_classify, mm_filenames, seq_filenames, show_ranked, codebook_size)


def nb_show(filename):
    check(lib.ecoz2_nb_show(str(filename).encode()))


def mm_show(filename):
    check(lib.ecoz2_mm_show(str(filename).encode()))


def nb_log_prob(nb_filename, seq_filename):
    out = C.c_double()
    check(lib.e2vq_nb_log_prob(str(nb_filename).encode(), str(seq_filename).encode(), C.byref(out)))
    return out.value


def mm_log_prob(mm_filename, seq_filename):
    out = C.c_float()
    check(lib.e2vq_mm_log_prob(str(mm_filename).encode(), str(seq_filename).encode(), C.byref(out)))
    return out.value


def c12n_run(model_class_names, class_ids, case_class_names, case_titles, probs, show_ranked, out_base_name):
    """C12nResults::add_case for every row of `probs`, then report_results; returns (result, confusion) tables"""
    import numpy as np

    n = len(model_class_names)
    probs = np.ascontiguousarray(probs, dtype=np.float64).reshape(len(class_ids), n)
    names, _k1 = _strs(model_class_names)
    ccn, _k2 = _strs(case_class_names)
    ttl, _k3 = _strs(case_titles)
    ids = (C.c_int * len(class_ids))(*class_ids)
    res = (C.c_int * ((n + 1) * (n + 1)))()
    conf = (C.c_int * ((n + 1) * (n + 1)))()
    check(lib.e2vq_c12n_run(names, n, ids, ccn, ttl, probs.ctypes.data_as(C.POINTER(C.c_double)), len(class_ids),
                            int(show_ranked), str(out_base_name).encode(), res, conf))
    return (np.array(res).reshape(n + 1, n + 1), np.array(conf).reshape(n + 1, n + 1))
