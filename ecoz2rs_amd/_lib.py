"""ctypes binding of libecoz2vq.so (declared in include/ecoz2_vq.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.environ.get("ECOZ2VQ_LIB") or os.path.join(_HERE, "csrc", "libecoz2vq.so")  # override: A/B builds


class Ecoz2Error(RuntimeError):
    pass


if not os.path.exists(lib_path):
    raise ImportError(
        f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C ecoz2rs_amd/csrc` (there is no pure-Python / CPU implementation)"
    )

lib = C.CDLL(lib_path)

c_double_p = C.POINTER(C.c_double)
c_char_pp = C.POINTER(C.c_char_p)

LEARN_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


class LevelStatsC(C.Structure):
    _fields_ = [
        ("M", C.c_int),
        ("passes", C.c_int),
        ("DD", C.c_double),
        ("avg_distortion", C.c_double),
        ("sigma", C.c_double),
        ("inertia", C.c_double),
        ("empty_cells", C.c_int64),
        ("failed_cells", C.c_int64),
    ]


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# Part 1: the reference's FFI surface (src/ecoz2_lib/mod.rs:72-178)
_sig("ecoz2_version", C.c_char_p)
_sig("ecoz2_vq_learn", C.c_int, C.c_int, C.c_double, C.c_char_p, c_char_pp, C.c_int, C.c_void_p, LEARN_CALLBACK)
_sig("ecoz2_vq_learn_using_base_codebook", C.c_int, C.c_char_p, C.c_double, c_char_pp, C.c_int, C.c_void_p,
     LEARN_CALLBACK)
_sig("ecoz2_vq_quantize", C.c_int, C.c_char_p, c_char_pp, C.c_int, C.c_int)
_sig("ecoz2_vq_show", C.c_int, C.c_char_p, C.c_int, C.c_int)
_sig("ecoz2_prd_show_file", C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int)
_sig("ecoz2_vq_classify", C.c_int, c_char_pp, C.c_int, c_char_pp, C.c_int, C.c_int)

# Part 2: session API
_sig("e2vq_last_error", C.c_char_p)
_sig("e2vq_device_count", C.c_int)
_sig("e2vq_session_create", C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p))
_sig("e2vq_session_destroy", None, C.c_void_p)
_sig("e2vq_set_stream", C.c_int, C.c_void_p, C.c_void_p)
_sig("e2vq_set_allreduce", C.c_int, C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int)
_sig("e2vq_enable_collective_timing", C.c_int, C.c_void_p, C.c_int)
_sig("e2vq_collective_timing", C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64))
_sig("e2vq_set_prefilter", C.c_int, C.c_void_p, C.c_int)
_sig("e2vq_group_create", C.c_int, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_void_p))
_sig("e2vq_group_bind", C.c_int, C.c_void_p, C.c_int, C.c_void_p)
_sig("e2vq_group_collective", C.c_char_p, C.c_void_p)
_sig("e2vq_group_uses_rccl", C.c_int, C.c_void_p)
_sig("e2vq_group_fail", None, C.c_void_p)
_sig("e2vq_group_destroy", None, C.c_void_p)
_sig("e2vq_set_frames_host", C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
_sig("e2vq_set_frames_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
_sig("e2vq_prepare", C.c_int, C.c_void_p)
_sig("e2vq_set_codebook", C.c_int, C.c_void_p, C.c_void_p, C.c_int)
_sig("e2vq_get_codebook", C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int))
_sig("e2vq_init_codebook", C.c_int, C.c_void_p)
_sig("e2vq_grow", C.c_int, C.c_void_p)
_sig("e2vq_pass", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("e2vq_pass_stats", C.c_int, C.c_void_p, C.POINTER(LevelStatsC))
_sig("e2vq_save_state", C.c_int, C.c_void_p)
_sig("e2vq_restore_state", C.c_int, C.c_void_p)
_sig("e2vq_verified_passes", C.c_int, C.c_void_p, C.POINTER(C.c_int64))
_sig("e2vq_update", C.c_int, C.c_void_p)
_sig("e2vq_enable_timing", C.c_int, C.c_void_p, C.c_int)
_sig("e2vq_last_pass_kernel_ms", C.c_int, C.c_void_p, C.POINTER(C.c_float))
_sig("e2vq_timing_total", C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64))
_sig("e2vq_timing_sweep_total", C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64))
_sig("e2vq_iterate", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LevelStatsC))
_sig("e2vq_last_pass_info", C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64))
_sig("e2vq_last_pass_records", C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64))
_sig("e2vq_last_pass_sweep", C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double))
_sig("e2vq_sweep_executed", C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int)
_sig("e2vq_set_sweep_policy", C.c_int, C.c_void_p, C.c_double, C.c_double)
_sig("e2vq_sweep_policy_state", C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64))
_sig("e2vq_set_prev_distortion", C.c_int, C.c_void_p, C.c_double)
_sig("e2vq_get_prev_distortion", C.c_int, C.c_void_p, C.POINTER(C.c_double))
_sig("e2vq_sweep_launch_counts", C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
_sig("e2vq_launch_counts_by_kernel", C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64))
_sig("e2vq_row_stride", C.c_int, C.c_int)
_sig("e2vq_get_rows", C.c_int, C.c_void_p, C.c_void_p)
_sig("e2vq_learn", C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p, LEARN_CALLBACK,
     C.POINTER(LevelStatsC), C.c_int, C.POINTER(C.c_int))
_sig("e2vq_quantize_host", C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
_sig("e2vq_quantize_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
_sig("e2vq_synchronize", C.c_int, C.c_void_p)
_sig("e2vq_avg_distortion_host", C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double))
_sig("e2vq_prd_info", C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int64))
_sig("e2vq_prd_read", C.c_int, C.c_char_p, C.c_void_p, C.c_int64)
_sig("e2vq_prd_write", C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64)
_sig("e2vq_cbook_info", C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int))
_sig("e2vq_cbook_read", C.c_int, C.c_char_p, C.c_void_p, C.c_int)
_sig("e2vq_cbook_write", C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p)
_sig("e2vq_seq_write", C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64)
_sig("e2vq_synth_frames", C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p)
_sig("e2vq_synth_frames_kind", C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int64, C.c_int64, C.c_void_p)


def check(rc):
    if rc != 0:
        raise Ecoz2Error(lib.e2vq_last_error().decode(errors="replace"))
