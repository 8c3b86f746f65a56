"""Host-side mirror of the reference's VQ wrappers.

``vq_learn`` / ``vq_quantize`` / ``vq_show`` take the arguments of the Rust functions of the same
name (/root/reference/src/ecoz2_lib/mod.rs:252-368) and call the same C symbols.
``VqSession`` wraps the resident-data session API (include/ecoz2_vq.h, part 2).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import ALLREDUCE_FN, LEARN_CALLBACK, LevelStatsC, c_char_pp, check, lib


def version():
    return lib.ecoz2_version().decode()


def _to_vec_of_ptr_const_c_char(paths):
    # to_vec_of_ptr_const_c_char, src/ecoz2_lib/mod.rs:530-543
    arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
    return C.cast(arr, c_char_pp), arr


def _observer(callback):
    def step(_target, m, avg_distortion, sigma, inertia):
        # Ecoz2ObserverRef::step, src/ecoz2_lib/mod.rs:61-69
        print(f"   Ecoz2ObserverRef.step: M={m} avg_distortion={avg_distortion} sigma={sigma} inertia={inertia}")
        if callback is not None:
            callback(m, avg_distortion, sigma, inertia)

    return LEARN_CALLBACK(step)


def vq_learn(base_codebook_opt, prediction_order_opt, epsilon, codebook_class_name, predictor_filenames,
             exp_key=None, callback=None):
    """ecoz2_lib::vq_learn (src/ecoz2_lib/mod.rs:252-323). ``exp_key`` is accepted and ignored."""
    assert (base_codebook_opt is not None) != (prediction_order_opt is not None)
    print(f"vq_learn: base_codebook_opt={base_codebook_opt!r} prediction_order={prediction_order_opt!r}, "
          f"epsilon={epsilon} codebook_class_name={codebook_class_name} "
          f"predictor_filenames: {len(predictor_filenames)}")
    files, _keep = _to_vec_of_ptr_const_c_char(predictor_filenames)
    cb = _observer(callback)
    if base_codebook_opt is not None:
        rc = lib.ecoz2_vq_learn_using_base_codebook(str(base_codebook_opt).encode(), float(epsilon), files,
                                                    len(predictor_filenames), None, cb)
    else:
        rc = lib.ecoz2_vq_learn(int(prediction_order_opt), float(epsilon), codebook_class_name.encode(), files,
                                len(predictor_filenames), None, cb)
    check(rc)


def vq_quantize(nom_raas, predictor_filenames, show_filenames=False):
    """ecoz2_lib::vq_quantize (src/ecoz2_lib/mod.rs:325-342)."""
    print(f"nom_raas = {nom_raas}")
    files, _keep = _to_vec_of_ptr_const_c_char(predictor_filenames)
    check(lib.ecoz2_vq_quantize(str(nom_raas).encode(), files, len(predictor_filenames), int(show_filenames)))


def vq_classify(cb_filenames, prd_filenames, show_ranked=False):
    """ecoz2_lib::vq_classify (src/ecoz2_lib/mod.rs:344-358)."""
    cbs, _k1 = _to_vec_of_ptr_const_c_char(cb_filenames)
    prds, _k2 = _to_vec_of_ptr_const_c_char(prd_filenames)
    check(lib.ecoz2_vq_classify(cbs, len(cb_filenames), prds, len(prd_filenames), int(show_ranked)))


def vq_show(codebook_filename, from_=-1, to=-1):
    """ecoz2_lib::vq_show (src/ecoz2_lib/mod.rs:360-368)."""
    print(f"codebook_filename = {codebook_filename}")
    check(lib.ecoz2_vq_show(str(codebook_filename).encode(), int(from_), int(to)))


def prd_show_file(prd_filename, show_reflections=False, from_=1, to=0):
    """ecoz2_lib::prd_show_file (src/ecoz2_lib/mod.rs:227-239; caller src/prd/mod.rs:99)."""
    check(lib.ecoz2_prd_show_file(str(prd_filename).encode(), int(show_reflections), int(from_), int(to)))


@dataclass
class LevelStats:
    M: int
    passes: int
    DD: float
    avg_distortion: float
    sigma: float
    inertia: float
    empty_cells: int
    failed_cells: int

    @classmethod
    def from_c(cls, c):
        return cls(c.M, c.passes, c.DD, c.avg_distortion, c.sigma, c.inertia, c.empty_cells, c.failed_cells)


def _ptr(x):
    """Device pointer of a torch tensor / int address / None."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class VqGroup:
    """The library's in-process group (what ecoz2_vq_learn builds for ECOZ2_VQ_GPUS > 1): rank r on devices[r], the
    exchange inside the library -- RCCL (one device per rank) or its peer-to-peer kernel.  Drive one VqSession per rank
    from one thread per rank (ctypes releases the GIL during the calls): every rank makes the same calls."""

    def __init__(self, devices, collective=None):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        check(lib.e2vq_group_create(len(self.devices), arr, collective.encode() if collective else None, C.byref(self._h)))

    def bind(self, rank, session):
        check(lib.e2vq_group_bind(self._h, int(rank), session._h))

    @property
    def collective(self):
        return lib.e2vq_group_collective(self._h).decode()

    @property
    def uses_rccl(self):
        return bool(lib.e2vq_group_uses_rccl(self._h))

    def fail(self):
        lib.e2vq_group_fail(self._h)

    def close(self):
        if self._h:
            lib.e2vq_group_destroy(self._h)
            self._h = C.c_void_p()


class VqSession:
    """One resident training set + codebook on one GPU (one rank of a sharded run)."""

    def __init__(self, prediction_order, device=0):
        self.P = int(prediction_order)
        self._h = C.c_void_p()
        check(lib.e2vq_session_create(int(device), self.P, C.byref(self._h)))
        self._ar = None

    def close(self):
        if self._h:
            lib.e2vq_session_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing -------------------------------------------------------------------
    def set_stream(self, hip_stream):
        check(lib.e2vq_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_allreduce(self, fn, rank, world):
        """fn(device_ptr:int, count:int, op:int, stream:int) -> None; op 0 = int64 sum, 1 = uint64 max."""

        def tramp(_user, buf, count, op, stream):
            try:
                fn(int(buf or 0), int(count), int(op), int(stream or 0))
                return 0
            except Exception as e:  # never unwind across the C boundary
                print(f"all-reduce hook raised: {e!r}")
                return 1

        self._ar = ALLREDUCE_FN(tramp)
        check(lib.e2vq_set_allreduce(self._h, self._ar, None, int(rank), int(world)))

    def synchronize(self):
        check(lib.e2vq_synchronize(self._h))

    def enable_collective_timing(self, on=True):
        check(lib.e2vq_enable_collective_timing(self._h, int(on)))

    def collective_timing(self):
        """(device ms between the events around the all-reduce hook, calls, bytes) since enable_collective_timing(True)"""
        ms, n, b = C.c_double(), C.c_int64(), C.c_int64()
        check(lib.e2vq_collective_timing(self._h, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def set_prefilter(self, on):
        """False: every pass on the plain FP64 sweep; True: the prefiltered sweep again (same results)"""
        check(lib.e2vq_set_prefilter(self._h, int(bool(on))))

    # -- training set -----------------------------------------------------------------
    def set_frames(self, frames):
        """frames: (T, P+1) float64 numpy array, or a CUDA/HIP torch tensor of that shape."""
        if hasattr(frames, "data_ptr"):
            assert frames.is_contiguous() and frames.shape[1] == self.P + 1
            import torch

            torch.cuda.current_stream(frames.device).synchronize()  # the producer of `frames` has finished
            check(lib.e2vq_set_frames_device(self._h, frames.data_ptr(), frames.shape[0]))
        else:
            a = np.ascontiguousarray(frames, dtype=np.float64)
            assert a.ndim == 2 and a.shape[1] == self.P + 1
            check(lib.e2vq_set_frames_host(self._h, a.ctypes.data, a.shape[0]))

    def prepare(self):
        check(lib.e2vq_prepare(self._h))

    # -- codebook ---------------------------------------------------------------------
    def set_codebook(self, reflections):
        a = np.ascontiguousarray(reflections, dtype=np.float64)
        assert a.ndim == 2 and a.shape[1] == self.P + 1
        check(lib.e2vq_set_codebook(self._h, a.ctypes.data, a.shape[0]))

    def codebook_size(self):
        m = C.c_int()
        check(lib.e2vq_get_codebook(self._h, None, C.byref(m)))
        return m.value

    def get_codebook(self):
        out = np.empty((self.codebook_size(), self.P + 1), dtype=np.float64)
        check(lib.e2vq_get_codebook(self._h, out.ctypes.data, None))
        return out

    def init_codebook(self):
        check(lib.e2vq_init_codebook(self._h))

    def grow(self):
        check(lib.e2vq_grow(self._h))

    # -- LBG iteration ------------------------------------------------------------------
    def run_pass(self, device_sym=None, device_dmin=None):
        check(lib.e2vq_pass(self._h, _ptr(device_sym), _ptr(device_dmin)))

    def pass_stats(self):
        st = LevelStatsC()
        check(lib.e2vq_pass_stats(self._h, C.byref(st)))
        return LevelStats.from_c(st)

    def update(self):
        check(lib.e2vq_update(self._h))

    def iterate(self, device_sym=None, device_dmin=None):
        """One whole LBG iteration (pass + statistics + centroid update) in one library call."""
        st = LevelStatsC()
        check(lib.e2vq_iterate(self._h, _ptr(device_sym), _ptr(device_dmin), C.byref(st)))
        return LevelStats.from_c(st)

    def enable_timing(self, on=True):
        check(lib.e2vq_enable_timing(self._h, int(on)))

    def last_pass_kernel_ms(self):
        ms = C.c_float()
        check(lib.e2vq_last_pass_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def timing_total(self):
        """(sum of the event-measured sweep-kernel times in ms, number of passes) since enable_timing(True)."""
        ms, n = C.c_double(), C.c_int64()
        check(lib.e2vq_timing_total(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_sweep_total(self):
        """The same for the sweep kernels alone (a pass with a separate accumulate kernel counts with both in timing_total)."""
        ms, n = C.c_double(), C.c_int64()
        check(lib.e2vq_timing_sweep_total(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_pass_records(self):
        """(did the last run_pass record its contributions for k_reduce_records?, records of the level's last recorded pass)."""
        r, n = C.c_int(), C.c_int64()
        check(lib.e2vq_last_pass_records(self._h, C.byref(r), C.byref(n)))
        return bool(r.value), n.value

    def last_pass_sweep(self):
        """(kind: 0 plain, 1 round-4 fused kernel, 2 candidate sweep + finish + reduce, 3 fused pass over grouped frames;
        two-stage?; flagged fraction of the level's first two-stage pass or -1)"""
        k, t, f = C.c_int(), C.c_int(), C.c_double()
        check(lib.e2vq_last_pass_sweep(self._h, C.byref(k), C.byref(t), C.byref(f)))
        return k.value, bool(t.value), f.value

    def sweep_executed(self, reset=False):
        """(flagged jobs, two-stage jobs, one-stage jobs) of the fused sorted passes since the last reset: e2vq_sweep_executed."""
        f, j, o = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.e2vq_sweep_executed(self._h, C.byref(f), C.byref(j), C.byref(o), int(reset)))
        return f.value, j.value, o.value

    def set_sweep_policy(self, two_stage_max_fraction=-1.0, max_uncertified_fraction=-1.0):
        """The host's two kernel switches (results never change): see e2vq_set_sweep_policy; negative = leave as it is."""
        check(lib.e2vq_set_sweep_policy(self._h, float(two_stage_max_fraction), float(max_uncertified_fraction)))

    def sweep_policy_state(self):
        """(one-stage sorted passes up to this M, plain sweep from this M on, uncertified frames of the last prefiltered pass)"""
        a, b, u = C.c_int(), C.c_int(), C.c_int64()
        check(lib.e2vq_sweep_policy_state(self._h, C.byref(a), C.byref(b), C.byref(u)))
        return a.value, b.value, u.value

    def last_pass_info(self):
        """(prefiltered sweep used?, frames it left to the full FP64 sweep) of the last run_pass."""
        used, n = C.c_int(), C.c_int64()
        check(lib.e2vq_last_pass_info(self._h, C.byref(used), C.byref(n)))
        return bool(used.value), n.value

    def set_prev_distortion(self, dd):
        """DDprv of the stopping rule (carries over between codebook sizes); see e2vq_set_prev_distortion."""
        check(lib.e2vq_set_prev_distortion(self._h, float(dd)))

    def save_state(self):
        """Save this point of the ladder (codebook, DDprv, rows and cells of the last pass): see e2vq_save_state."""
        check(lib.e2vq_save_state(self._h))

    def restore_state(self):
        check(lib.e2vq_restore_state(self._h))

    def prev_distortion(self):
        dd = C.c_double()
        check(lib.e2vq_get_prev_distortion(self._h, C.byref(dd)))
        return dd.value

    def sweep_launch_counts(self):
        """(prefiltered, plain) sweep launches of training passes so far"""
        a, b = C.c_int64(), C.c_int64()
        check(lib.e2vq_sweep_launch_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def launch_counts_by_kernel(self):
        """(k_pass_pre_lds, k_sweep_cand, plain) sweep launches of training passes so far"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.e2vq_launch_counts_by_kernel(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_rows(self):
        rs = lib.e2vq_row_stride(self.P)
        out = np.empty((self.codebook_size(), rs), dtype=np.int64)
        check(lib.e2vq_get_rows(self._h, out.ctypes.data))
        return out

    def learn(self, epsilon, max_M, class_name="_", out_root=None, callback=None):
        levels = (LevelStatsC * 32)()
        n = C.c_int()
        cb = LEARN_CALLBACK((lambda _t, m, a, s, i: callback(m, a, s, i)) if callback else (lambda *_: None))
        check(lib.e2vq_learn(self._h, float(epsilon), int(max_M), class_name.encode(),
                             out_root.encode() if out_root else None, None, cb, levels, 32, C.byref(n)))
        return [LevelStats.from_c(levels[i]) for i in range(min(n.value, 32))]

    # -- quantize -------------------------------------------------------------------------
    def quantize(self, frames, want_dmin=True):
        a = np.ascontiguousarray(frames, dtype=np.float64)
        assert a.ndim == 2 and a.shape[1] == self.P + 1
        sym = np.empty(a.shape[0], dtype=np.uint16)
        dmin = np.empty(a.shape[0], dtype=np.float64) if want_dmin else None
        check(lib.e2vq_quantize_host(self._h, a.ctypes.data, a.shape[0], sym.ctypes.data,
                                     dmin.ctypes.data if want_dmin else None))
        return (sym, dmin) if want_dmin else sym

    def avg_distortion(self, frames):
        a = np.ascontiguousarray(frames, dtype=np.float64)
        out = C.c_double()
        check(lib.e2vq_avg_distortion_host(self._h, a.ctypes.data, a.shape[0], C.byref(out)))
        return out.value

    def quantize_device(self, device_frames, T, device_sym, device_dmin=None):
        check(lib.e2vq_quantize_device(self._h, _ptr(device_frames), int(T), _ptr(device_sym), _ptr(device_dmin)))
