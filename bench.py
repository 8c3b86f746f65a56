#!/usr/bin/env python3
"""bench.py -- VQ-learn frames/s at M=1024, P=36 on N MI355X (BASELINE.json metric).

A "step" is one LBG pass of the REAL M=1024 level over the resident shard of every rank:
  sweep+accumulate kernel (K1+K2)  ->  int64 all-reduce of the cell sums (RCCL, N>1)
  ->  level statistics (the host reads DD for the convergence test)  ->  centroid update (K3/K4).
The timed region repeats the level as the ladder runs it: restore the point where the M=512 level ended (device copies
of the codebook, its rows and cells, and the rebuild of the codeword images: inside the timed region), split to M=1024,
then passes until (DDprv-DD)/DD < eps ends the level (3 passes on this data: the first seeded with the parents' sums,
two incremental ones), through the library's own e2vq_learn.  K steps = K such passes (whole
levels; a remainder of K is run as the leading passes of one more level).
Frames are synthetic (seeded, counter based: rank r holds frames [r*S, (r+1)*S) of one stream) and
resident in HBM before the timed region.  Weak scaling: S = 2^21 frames per GPU (config 4's shard).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...       (no launcher: the parent starts the N ranks itself as fresh child processes,
                                        before it has made any GPU call, and relays rank 0's JSON line)
    python bench.py --gpus N --in-process   the N ranks as the library's OWN in-process group (e2vq_group_*: what
                                        ecoz2_vq_learn runs for ECOZ2_VQ_GPUS=N behind the reference's single-process
                                        caller), one host thread per rank: the exchange timed is the library's --
                                        ncclAllReduce(int64) on RCCL loaded by the library when every rank has a GPU of its
                                        own, its peer-to-peer kernel when ranks share one -- not torch.distributed's
After the timed region (untimed) the line's `config.parity` is made: the same level re-run on the plain FP64 sweep must give
the same codebook bit for bit, and the per-frame outputs of the timed kernel are checked against the strict CPU oracle on a
32 768-frame sample; a mismatch makes the run fail.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

P = 36
M = 1024
FRAMES_PER_GPU = 1 << 21
SEED = 20244  # 20240 + config# 
N_CLASSES = 20
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 peak (vector == matrix on CDNA4), AMD spec
F16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense BF16/F16 MFMA peak ~2.5 PF (the prefilter's limb products)
F16_MFMA_FLOP_PER_FRAME_CODEWORD = 2 * 16 * 15  # 15 k-steps of v_mfma_f32_32x32x16_f16 per (frame, codeword) pair
BYTES_PER_FRAME_PASS = 306  # 296 B frame + 2 B symbol + 8 B min distortion
FLOP_PER_FRAME_PASS = 2 * M * (P + 1)


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the device/host sources of the library: ties a PMC traffic figure to a build"""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "ecoz2rs_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def cpu_quota_cores():
    """CPU share of this container (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(e, np):
    """Reference-flags CPU port (oracle source, -O3 -ffast-math -fopenmp) on this host's cores."""
    from tests import oracle_lib

    variant = "libvqoracle_fast.so"
    try:  # rebuild for this host's ISA (build.rs uses -march=native)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        variant = "libvqoracle_fast_native.so"
    except Exception:
        pass
    fast = oracle_lib.load(variant)
    strict = oracle_lib.load()
    frames = e.synth.synth_frames(SEED, N_CLASSES, P, 0, 1 << 15)
    # a plausible M=1024 codebook: reflections of 1024 of the frames
    refl = np.zeros((M, P + 1))
    for i in range(M):
        _st, _pe, rc, _a = strict.lpca_r(frames[i * 7], P)
        refl[i, 1:] = rc[1:]
    cq = strict.reflections_to_cq(refl)
    # the box's CPU share is smaller than the visible core count: pick the thread count that runs fastest
    ncpu = len(os.sched_getaffinity(0))
    best_rate, best_nt = 0.0, 1
    for nt in sorted({1, 8, 16, 32, 64, ncpu} & set(range(1, ncpu + 1))):
        fast.set_threads(nt)
        secs, _ = fast.time_pass(cq, frames[:8192], 1)
        if 8192 / secs > best_rate:
            best_rate, best_nt = 8192 / secs, nt
    fast.set_threads(best_nt)
    rate = best_rate
    n = int(min(frames.shape[0], max(4096, rate * 1.5)))
    reps = max(1, int(round(12.0 * rate / n)))
    secs, threads = fast.time_pass(cq, frames[:n], reps)
    return {
        "value": n * reps / secs,
        "unit": "frames/s",
        "cores": threads,
        "per_thread": n * reps / secs / max(1, threads),
        "gflops": n * reps / secs * FLOP_PER_FRAME_PASS / 1e9,
        "visible_cores": ncpu,
        "cpu_quota_cores": cpu_quota_cores(),
        "caveat": "stand-in for the unbuildable reference C library (kind: port); assignment pass only (the accumulate "
                  "adds 37 of 37 888 operations per frame); thread count picked by calibration because the box's CPU "
                  "quota can be below its visible cores; a GPU/CPU ratio is not a measure of kernel quality",
        "kind": "port",
        "sample": f"{reps} assignment passes over {n} synthetic frames at M={M}, P={P} "
                  f"({secs:.1f} s; oracle source built with the reference's flags -O3 -ffast-math -fopenmp, {variant})",
    }


SMALL_CORPUS_T = 38265     # the one vq-learn run the reference documents: notes.md:122-153 (38 265 vectors, eps 0.05, M = 2 ... 2048)
SMALL_CORPUS_M = 2048


def small_corpus(e, np, with_cpu):
    """Untimed extra (VERDICT r04 task 5): the workload a drop-in user runs first.  `ecoz2 vq learn` from a .prd file as a
    fresh process (HIP start-up included) and warm through the same C entry point, wall time to the M = 2048 .cbook; the ladder
    level by level on resident frames (passes, kernel and step time per pass: all launch latency at this size); `ecoz2 vq
    quantize` of the same corpus split into ~100-frame files; and -- rank 0, next to it -- the CPU stand-in (the oracle source
    with the reference's flags, all host threads) on the same frames."""
    import shutil
    import tempfile

    T, MM = SMALL_CORPUS_T, SMALL_CORPUS_M
    root = tempfile.mkdtemp(prefix="e2small_")
    out = {"what": f"notes.md:122-153: {T} training vectors, eps 0.05, M = 2 ... {MM}, P = {P}; synthetic frames (seed {SEED})",
           "frames": T, "max_codebook_size": MM}
    try:
        frames = e.synth.synth_frames(SEED, N_CLASSES, P, 0, T)
        prd = os.path.join(root, "data", "predictors", "_", "corpus.prd")
        e.formats.write_prd(prd, "_", frames)
        env = dict(os.environ, ECOZ2_VQ_OUT_ROOT=root, ECOZ2_VQ_MAX_CODEBOOK_SIZE=str(MM), ECOZ2_VQ_QUIET="1")
        cli = os.path.join(ROOT, "ecoz2rs_amd", "csrc", "ecoz2")
        cold = []
        for _ in range(3):  # a fresh process each (a child of this one; nothing is exec'ed over a GPU process)
            shutil.rmtree(os.path.join(root, "data", "codebooks"), ignore_errors=True)
            t0 = time.perf_counter()
            subprocess.run([cli, "vq", "learn", "-P", str(P), "-e", "0.05", "--predictors", prd], env=env, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            cold.append(time.perf_counter() - t0)
        cb = os.path.join(root, "data", "codebooks", "_", f"eps_0.05_M_{MM:04d}.cbook")
        out["cli_cold_seconds"] = [round(x, 4) for x in cold]
        out["cli_wrote_codebook"] = os.path.exists(cb)
        # warm: the reference's entry point in this process (device and library initialised)
        saved = {k: os.environ.get(k) for k in ("ECOZ2_VQ_OUT_ROOT", "ECOZ2_VQ_MAX_CODEBOOK_SIZE")}
        os.environ.update(ECOZ2_VQ_OUT_ROOT=root, ECOZ2_VQ_MAX_CODEBOOK_SIZE=str(MM))
        warm = []
        try:
            import contextlib
            import io

            for _ in range(3):
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    e.vq_learn(None, P, 0.05, "_", [prd])
                warm.append(time.perf_counter() - t0)
            out["entry_point_warm_seconds"] = [round(x, 4) for x in warm]
            # quantize: the corpus as ~100-frame files (383 files), cold CLI and warm entry point
            small = []
            for i in range(0, T, 100):
                f = os.path.join(root, "data", "predictors", "q", f"{i // 100:05d}.prd")
                e.formats.write_prd(f, "q", frames[i:i + 100])
                small.append(f)
            t0 = time.perf_counter()
            subprocess.run([cli, "vq", "quantize", "--codebook", cb, "--predictors", os.path.join(root, "data", "predictors", "q")],
                           env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            out["quantize_cli_cold_seconds"] = round(time.perf_counter() - t0, 4)
            qw = []
            for _ in range(3):
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    e.vq_quantize(cb, small, False)
                qw.append(time.perf_counter() - t0)
            out["quantize_files"] = len(small)
            out["quantize_entry_point_warm_seconds"] = [round(x, 4) for x in qw]
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        # the ladder on resident frames, level by level (one synchronisation per level)
        levels = []
        with e.VqSession(P, device=0) as s:
            s.set_frames(frames)
            s.prepare()
            for rep in range(2):  # (the second repetition is the one reported: every kernel loaded)
                levels = []
                s.init_codebook()
                s.synchronize()
                t_all = time.perf_counter()
                m = 2
                while m <= MM:
                    s.enable_timing(True)
                    s.synchronize()
                    t0 = time.perf_counter()
                    lv = s.learn(0.05, m)[0]
                    s.synchronize()
                    wall = time.perf_counter() - t0
                    kms, kn = s.timing_total()
                    levels.append({"M": m, "passes": lv.passes, "kernel_us_per_pass": round(1e3 * kms / max(1, kn), 1),
                                   "step_us_per_pass": round(1e6 * wall / lv.passes, 1)})
                    m *= 2
                t_all = time.perf_counter() - t_all
                s.enable_timing(False)
            s.init_codebook()
            s.synchronize()
            t0 = time.perf_counter()
            whole = s.learn(0.05, MM)
            s.synchronize()
            out["resident_ladder_seconds"] = round(time.perf_counter() - t0, 5)
            out["resident_ladder_passes"] = sum(x.passes for x in whole)
            out["resident_ladder_us_per_pass"] = round(1e6 * (time.perf_counter() - t0) / max(1, out["resident_ladder_passes"]), 1)
        out["levels"] = levels
        if with_cpu:
            from tests import oracle_lib

            variant = "libvqoracle_fast_native.so" if os.path.exists(os.path.join(ROOT, "oracle", "_build", "libvqoracle_fast_native.so")) \
                else "libvqoracle_fast.so"
            fast = oracle_lib.load(variant)
            best = None
            for nt in (8, 16, 32):
                fast.set_threads(nt)
                t0 = time.perf_counter()
                rc, lv_o, _cbs = fast.learn(frames, 0.05, MM)
                dt = time.perf_counter() - t0
                if rc == 0 and (best is None or dt < best[0]):
                    best = (dt, nt, sum(x["passes"] for x in lv_o))
            if best:
                out["cpu_stand_in"] = {"seconds": round(best[0], 4), "threads": best[1], "passes": best[2], "kind": "port",
                                       "what": f"the oracle source with the reference's flags ({variant}), whole ladder in memory "
                                               "(no file I/O), fastest of 8 / 16 / 32 threads"}
    except Exception as ex:  # informational: never fails the bench line
        out["error"] = repr(ex)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (never an exec; this
    parent has made no GPU call), one per GPU -- LOCAL_RANK modulo the device count inside the child, so that a
    1-GPU box can rehearse N ranks with --backend gloo --, wait for all of them, relay rank 0's stdout (the JSON line)
    and exit non-zero if any rank did."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py: rank(s) failed: {bad}")



class ProcComm:
    """one process per GPU (the driver's launch, or self_launch): ranks meet through torch.distributed"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.args = torch, dist, args
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        self.local = local % torch.cuda.device_count()
        torch.cuda.set_device(self.local)
        if self.world > 1:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{self.local}"))
            else:
                dist.init_process_group(args.backend)
        elif args.force_collective:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(f"cuda:{self.local}"))
            os.environ["ECOZ2_VQ_FORCE_ALLREDUCE"] = "1"
        self.ar_calls = {"n": 0, "bytes": 0, "max_bytes": 0}

    def attach(self, sess):
        from ecoz2rs_amd import parallel

        parallel.bind_torch_stream(sess, self.local)  # session kernels + RCCL collectives on one torch stream
        if self.world > 1 or self.args.force_collective:
            inner = parallel.make_allreduce(self.local)
            ar = self.ar_calls

            def counted(ptr, count, op, stream):
                ar["n"] += 1
                ar["bytes"] += 8 * count
                ar["max_bytes"] = max(ar["max_bytes"], 8 * count)
                inner(ptr, count, op, stream)

            sess.set_allreduce(counted, self.rank, self.world)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=f"cuda:{self.local}")
        if self.args.backend != "nccl":
            t = t.cpu()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def describe(self, ar_timed):
        if self.world == 1 and not self.args.force_collective:
            return None
        if self.world == 1:
            return {"backend": "nccl (one-rank group on this GPU: the host side of the exchange, no wire)", "world_size": 1,
                    "exchange": "torch.distributed all_reduce through the session's hook",
                    "allreduce_calls": ar_timed["n"], "allreduce_calls_per_step": ar_timed["n"] / max(1, self.args.steps),
                    "bytes_per_call": self.ar_calls["max_bytes"], "bytes_timed_region": ar_timed["bytes"]}
        devs = [None] * self.world
        self.dist.all_gather_object(devs, self.local)
        try:
            rccl = ".".join(str(x) for x in self.torch.cuda.nccl.version()) if self.args.backend == "nccl" else None
        except Exception:
            rccl = None
        return {
            "backend": self.args.backend + (" (RCCL over xGMI)" if self.args.backend == "nccl" else " (host-staged rehearsal)"),
            "exchange": "torch.distributed all_reduce through the session's hook (process per GPU)",
            "world_size": self.world,
            "devices_per_rank": 1,
            "device_of_rank": devs,
            "distinct_devices": len(set(devs)),
            "allreduce_calls": ar_timed["n"],
            "allreduce_calls_per_step": ar_timed["n"] / max(1, self.args.steps),
            "bytes_per_call": self.ar_calls["max_bytes"],
            "bytes_timed_region": ar_timed["bytes"],
            "dtype": "int64 sum (exact: any rank count gives the same bits)",
            "rccl_version": rccl,
        }

    def counters(self):
        return dict(self.ar_calls)

    def finish(self):
        if self.world > 1 or (self.world == 1 and self.args.force_collective):
            try:
                self.dist.destroy_process_group()
            except Exception:
                pass


class ThreadComm:
    """--in-process: rank r is a host thread driving its own session; the ranks meet in the library's group"""

    def __init__(self, args, shared, rank):
        import torch

        self.torch, self.args, self.shared = torch, args, shared
        self.rank, self.world = rank, args.gpus
        self.local = shared["devices"][rank]
        torch.cuda.set_device(self.local)  # (per thread)

    def attach(self, sess):
        self.shared["group"].bind(self.rank, sess)

    def barrier(self):
        self.shared["barrier"].wait()

    def max_over_ranks(self, x):
        sl = self.shared["slots"]
        sl[self.rank] = x
        self.shared["barrier"].wait()
        m = max(sl)
        self.shared["barrier"].wait()
        return m

    def describe(self, ar_timed):
        g = self.shared["group"]
        return {
            "backend": "rccl inside the library (dlopen, ncclCommInitAll, ncclAllReduce)" if g.uses_rccl
                       else "the library's peer-to-peer reduce-scatter + all-gather kernel (ranks share devices, or RCCL is not used)",
            "exchange": "the library's own in-process group (e2vq_group_*): the exchange ecoz2_vq_learn runs for ECOZ2_VQ_GPUS=N",
            "library_says": g.collective,
            "world_size": self.world,
            "devices_per_rank": 1,
            "device_of_rank": list(self.shared["devices"]),
            "distinct_devices": len(set(self.shared["devices"])),
            "dtype": "int64 sum (exact: any rank count gives the same bits)",
        }

    def counters(self):
        return {"n": 0, "bytes": 0, "max_bytes": 0}

    def finish(self):
        pass


def run_in_process(args):
    """the N ranks as threads of this process, meeting in the library's own group"""
    import threading

    import torch

    # (torch first: its bundled HIP runtime must initialise before the library's first HIP call, or it finds no GPU -- the
    # order the process-per-GPU path has always had; torch only provides the output buffers of the parity check here)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.init()
    import ecoz2rs_amd as e

    ndev = torch.cuda.device_count()
    devices = [r % ndev for r in range(args.gpus)]
    group = e.VqGroup(devices, args.collective or None)
    shared = {"devices": devices, "group": group, "barrier": threading.Barrier(args.gpus), "slots": [0.0] * args.gpus}
    errors = [None] * args.gpus

    def rank_main(r):
        try:
            run(args, ThreadComm(args, shared, r))
        except BaseException as ex:  # a rank that gives up releases the others from the library's rendezvous and from ours
            errors[r] = ex
            group.fail()
            shared["barrier"].abort()

    threads = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(1, args.gpus)]
    for t in threads:
        t.start()
    rank_main(0)
    for t in threads:
        t.join()
    group.close()
    bad = [(r, repr(ex)) for r, ex in enumerate(errors) if ex is not None]
    if bad:
        raise SystemExit(f"bench.py --in-process: rank(s) failed: {bad}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=FRAMES_PER_GPU)
    ap.add_argument("--codebook-size", type=int, default=1024,
                    help="time this level of the ladder instead of M = 1024 (profiles of the M = 256 / 512 kernels; the "
                         "metric of BASELINE.json is the default)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational extras (steady state, end-to-end ladder, quantize, 16 M-frame run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefilter", action="store_true",
                    help="keep every pass on the plain FP64 sweep (A/B against the prefiltered sweep; same results)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: a one-rank nccl group and the all-reduce hook on every pass anyway (what the exchange's host "
                         "side costs per pass, without a wire: DESIGN.md section 5)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL over xGMI; "
                    "gloo lets several ranks share one GPU when rehearsing the N > 1 path)")
    ap.add_argument("--in-process", action="store_true",
                    help="the N ranks as the library's own in-process group (one host thread per rank; rank r on GPU r modulo "
                         "the device count): times the library's exchange, not torch.distributed's")
    ap.add_argument("--collective", default="", help="--in-process: rccl | p2p (default: RCCL when every rank has a GPU of "
                    "its own, else the library's peer-to-peer kernel)")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed parity section (config.parity)")
    args = ap.parse_args()
    M = args.codebook_size
    if M < 4 or M & (M - 1):
        raise SystemExit("--codebook-size must be a power of two >= 4")
    if args.no_prefilter:
        os.environ["ECOZ2_VQ_PREFILTER"] = "0"
    if args.in_process and "WORLD_SIZE" not in os.environ:
        return run_in_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    comm = ProcComm(args)
    try:
        run(args, comm)
    finally:
        comm.finish()


def parity_section(args, comm, sess, whole_level, sym, dmin, S, lo, M):
    """Untimed.  (1) The level just timed, re-run with the prefilter off in the same session: codebook bytes, pass count and DD
    must be identical.  (2) One more pass of the timed kernel on the level's final codebook with per-frame outputs: symbols and
    distortions of a 32 768-frame sample against the strict CPU oracle (the checker; it is never the thing measured), and --
    one rank -- every cell's frame count against the symbols.  Raises on any mismatch."""
    import hashlib

    import numpy as np

    import ecoz2rs_amd as e

    lv_pre = whole_level()
    cb_pre = sess.get_codebook()
    prefiltered, _ = sess.last_pass_info()
    out = {"level": M, "passes": lv_pre.passes, "DD": lv_pre.DD.hex(),
           "codebook_sha16": hashlib.sha256(cb_pre.tobytes()).hexdigest()[:16], "timed_kernel_prefiltered": bool(prefiltered)}
    # (2) first: the session still holds the level's last pass
    sess.run_pass(sym, dmin)
    sess.synchronize()
    n = min(S, 32768)
    sym_h = sym[:n].cpu().numpy().view(np.uint16)
    dmin_h = dmin[:n].cpu().numpy()
    if comm.rank == 0:
        from tests import oracle_lib

        oracle = oracle_lib.load()
        sample = e.synth.synth_frames(SEED, N_CLASSES, P, lo, n)
        sym_o, dmin_o = oracle.quantize(oracle.reflections_to_cq(cb_pre), sample)
        bad = int(np.count_nonzero(sym_h != sym_o) + np.count_nonzero(dmin_h.view(np.uint64) != dmin_o.view(np.uint64)))
        out["oracle_sample_frames"] = n
        out["oracle_mismatches"] = bad
        if comm.world == 1:
            rows = sess.get_rows()
            counts = np.bincount(sym.cpu().numpy().view(np.uint16), minlength=M)
            out["cell_counts_match_symbols"] = bool(np.array_equal(rows[:, 2 * (P + 1)], counts))
    # (1) the same level on the plain FP64 sweep
    if prefiltered:
        sess.set_prefilter(False)
        lv_plain = whole_level()
        cb_plain = sess.get_codebook()
        plain_used, _ = sess.last_pass_info()
        sess.set_prefilter(True)
        out["equals_plain_sweep"] = bool(not plain_used and cb_plain.tobytes() == cb_pre.tobytes() and
                                         lv_plain.passes == lv_pre.passes and lv_plain.DD.hex() == lv_pre.DD.hex())
    else:
        out["equals_plain_sweep"] = None  # (the timed kernel IS the plain sweep)
    ok = out.get("oracle_mismatches", 0) == 0 and out["equals_plain_sweep"] is not False and \
        out.get("cell_counts_match_symbols", True)
    out["ok"] = bool(ok)
    return out


def run(args, comm):
    import numpy as np
    import torch

    import ecoz2rs_amd as e

    M = args.codebook_size
    FLOP_PER_FRAME_PASS = 2 * M * (P + 1)
    rank, world, local = comm.rank, comm.world, comm.local
    S = args.frames_per_gpu
    lo = rank * S
    frames = e.synth.synth_frames(SEED, N_CLASSES, P, lo, S)

    sess = e.VqSession(P, device=local)
    comm.attach(sess)
    sess.set_frames(frames)  # H2D + blocked re-layout; resident from here on
    del frames
    sess.prepare()
    sess.init_codebook()
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    t_ladder = time.time()
    levels = sess.learn(0.05, M // 2)  # real LBG ladder 2..512 (untimed) -> realistic codebook state
    sess.synchronize()
    t_ladder = time.time() - t_ladder
    sess.save_state()  # the converged M / 2 codebook, its DD, and the rows and cells the next level's seeded first pass starts from
    sym = torch.empty(S, dtype=torch.int16, device=f"cuda:{local}")
    dmin = torch.empty(S, dtype=torch.float64, device=f"cuda:{local}")
    EPS = 0.05

    def restore():
        sess.restore_state()

    def whole_level():
        """the M = 1024 level through the library's LBG driver (e2vq_learn): split + passes until convergence"""
        restore()
        return sess.learn(EPS, M)[0]

    def leading_passes(n):
        """the first n passes of the level, step by step (same calls, same order as e2vq_learn's loop)"""
        restore()
        sess.grow()
        for i in range(n):
            sess.run_pass()
            st_ = sess.pass_stats()
            if i + 1 < n:
                sess.update()
        return st_

    def fence():
        sess.synchronize()
        torch.cuda.synchronize(local)
        comm.barrier()
        sess.synchronize()

    # warm-up: whole levels (at least one: it also tells how many passes the level takes on this data)
    lv = whole_level()
    L = lv.passes
    done = L
    while done < args.warmup:
        whole_level()
        done += L
    sess.enable_timing(True)  # HIP events around the sweep kernel of every pass, summed inside the library
    sess.enable_collective_timing(True)  # ... and around every call of the exchange
    fence()
    launches_before = sess.launch_counts_by_kernel()  # (k_pass_pre_lds, k_sweep_cand, plain)
    ar_before = comm.counters()
    t0 = time.perf_counter()
    steps_left = args.steps
    while steps_left >= L:
        st = whole_level()
        assert st.passes == L
        steps_left -= L
    if steps_left:
        st = leading_passes(steps_left)
    fence()
    dt = time.perf_counter() - t0
    ar_now = comm.counters()
    ar_timed = {k: ar_now[k] - ar_before[k] for k in ("n", "bytes")}
    # the dominant kernel alone (the sweep), and the pass's kernels together (sweep + k_reduce_records where the accumulate
    # is a kernel of its own)
    kernel_ms_total, kernel_passes = sess.timing_sweep_total()
    pass_kernels_ms_total = sess.timing_total()[0]
    assert kernel_passes == args.steps, (kernel_passes, args.steps)
    ar_ms, ar_n, ar_b = sess.collective_timing()
    sess.enable_collective_timing(False)
    prefiltered, fallback_frames = sess.last_pass_info()
    launches_after = sess.launch_counts_by_kernel()
    timed_lds, timed_sweep, timed_plain = (launches_after[k] - launches_before[k] for k in range(3))
    timed_pre = timed_lds + timed_sweep
    sweep_kind, two_stage, flagged_frac = sess.last_pass_sweep()
    dt = comm.max_over_ranks(dt)
    collective = comm.describe(ar_timed)
    if collective is not None:
        # device time between the events the library records around its calls of the exchange (the collective's kernels and
        # their wait for the other ranks), this rank, timed region
        collective["allreduce_calls_timed"] = ar_n
        collective["allreduce_bytes_per_call"] = ar_b // max(1, ar_n)
        collective["allreduce_us_per_call"] = 1e3 * ar_ms / max(1, ar_n)
        collective["allreduce_ms_per_step"] = ar_ms / max(1, args.steps)

    parity = None
    if not args.no_parity:
        parity = parity_section(args, comm, sess, whole_level, sym, dmin, S, lo, M)

    extras = not args.no_extras
    steady_ms = steady_kernel_ms = e2e_s = q_rate = None
    e2e_levels, level_detail, big = [], [], None
    if extras:
        # ---- steady state (informational): back-to-back iterations on the converged codebook, where the
        # incremental accumulate has almost nothing left to move -- round 1's headline regime, kept as an extra key
        whole_level()
        for _ in range(3):
            sess.iterate(sym, dmin)
        sess.enable_timing(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(10):
            sess.iterate(sym, dmin)
        fence()
        steady_ms = (time.perf_counter() - t0) / 10 * 1e3
        steady_kernel_ms = sess.timing_total()[0] / 10

        # ---- secondary figures (SURVEY 8d ii / iii), outside the timed region, informational --------------------
        sess.enable_timing(False)
        sess.init_codebook()
        fence()
        t0 = time.perf_counter()
        e2e_levels = sess.learn(0.05, M)  # the whole ladder 2..M with the real convergence rule, one library call
        fence()
        e2e_s = time.perf_counter() - t0
        # the same ladder level by level (a synchronisation per level: ~30 us each), with the sweep kernel's event time:
        # per level {passes, kernel ms per pass, step ms per pass, what bounds the level's sweep, fraction of that bound}
        sess.init_codebook()
        m = 2
        while m <= M:
            sess.enable_timing(True)
            sess.enable_collective_timing(collective is not None)
            fence()
            t0 = time.perf_counter()
            lvm = sess.learn(0.05, m)[0]
            fence()
            wall = time.perf_counter() - t0
            kms, kn = sess.timing_total()
            lar_ms, lar_n, _ = sess.collective_timing()
            hbm_ms = BYTES_PER_FRAME_PASS * S / (HBM_PEAK_GBS * 1e9) * 1e3
            fp64_ms = 2.0 * m * (P + 1) * S / (FP64_PEAK_TFLOPS * 1e12) * 1e3
            f16_ms = F16_MFMA_FLOP_PER_FRAME_CODEWORD * m * S / (F16_PEAK_TFLOPS * 1e12) * 1e3
            lkind, ltwo, lfrac = sess.last_pass_sweep()
            if prefiltered and m >= 128 and lkind in (2, 3) and ltwo and lfrac >= 0:
                # two-stage sweep: 8 coarse k-steps for every (tile, column block), all 15 again for the flagged ones
                bound, bound_ms = "f16 mfma (executed limb products: two-stage sweep)", f16_ms * (8 + 15 * lfrac) / 15
            elif prefiltered and m >= 128:  # (the prefiltered pass serves M >= 128)
                bound, bound_ms = "f16 mfma (executed limb products)", f16_ms
            elif hbm_ms >= fp64_ms:
                bound, bound_ms = "hbm (306 B per frame-pass)", hbm_ms
            else:
                bound, bound_ms = "fp64 mfma (2 M (P+1) flop per frame-pass)", fp64_ms
            level_detail.append({"M": m, "passes": lvm.passes, "kernel_ms": kms / max(1, kn), "step_ms": wall / lvm.passes * 1e3,
                                 "bound": bound, "bound_ms": bound_ms, "frac_of_bound": bound_ms / (kms / max(1, kn)),
                                 "sweep_kind": lkind, "flagged_fraction": lfrac if ltwo else None})
            if collective is not None:
                level_detail[-1]["allreduce_us_per_call"] = 1e3 * lar_ms / max(1, lar_n)
                level_detail[-1]["allreduce_bytes"] = m * e.lib.e2vq_row_stride(P) * 8
            m *= 2
        sess.enable_timing(False)
        sess.enable_collective_timing(False)
        if world == 1:
            fr = torch.from_numpy(e.synth.synth_frames(SEED, N_CLASSES, P, lo, S)).cuda(local)
            for _ in range(3):
                torch.cuda.synchronize(local)
                t0 = time.perf_counter()
                sess.quantize_device(fr, S, sym, dmin)
                sess.synchronize()
                q_rate = max(q_rate or 0.0, S / (time.perf_counter() - t0))
            del fr
        # ---- config 4's whole set (2^24 frames) on ONE GPU: strong-scaling reference point, informational ----------------
        if world == 1 and M == 1024 and S == FRAMES_PER_GPU and not os.environ.get("ECOZ2_BENCH_SKIP_16M"):
            try:
                T16 = 1 << 24
                t0 = time.perf_counter()
                fr16 = e.synth.synth_frames(SEED, N_CLASSES, P, 0, T16)
                synth_s = time.perf_counter() - t0
                with e.VqSession(P, device=local) as s16:
                    t0 = time.perf_counter()
                    s16.set_frames(fr16)
                    s16.prepare()
                    s16.synchronize()
                    upload_s = time.perf_counter() - t0
                    del fr16
                    s16.init_codebook()
                    s16.learn(0.05, M // 2)
                    s16.enable_timing(True)
                    s16.synchronize()
                    t0 = time.perf_counter()
                    lv16 = s16.learn(0.05, M)[0]
                    s16.synchronize()
                    level_s = time.perf_counter() - t0
                    kms16, kn16 = s16.timing_total()
                    s16.enable_timing(False)
                    s16.init_codebook()
                    s16.synchronize()
                    t0 = time.perf_counter()
                    lad16 = s16.learn(0.05, M)
                    s16.synchronize()
                    ladder_s = time.perf_counter() - t0
                big = {
                    "what": f"config 4's whole training set ({T16} frames) resident on ONE GPU: the M={M} level and the whole ladder",
                    "frames": T16,
                    "host_synth_seconds": round(synth_s, 2),
                    "upload_relayout_statistics_seconds": round(upload_s, 3),
                    "level_passes": lv16.passes,
                    "level_kernel_ms_per_pass": kms16 / max(1, kn16),
                    "level_ms_per_pass": level_s / lv16.passes * 1e3,
                    "level_frames_per_sec": T16 * lv16.passes / level_s,
                    "ladder_seconds": round(ladder_s, 4),
                    "ladder_frames_per_sec": T16 / ladder_s,
                    "ladder_passes_per_level": [x.passes for x in lad16],
                }
            except Exception as ex:  # (a box short of host memory: the figure is informational)
                big = {"error": repr(ex)}

    small = None
    if extras and rank == 0 and world == 1 and M == 1024 and not os.environ.get("ECOZ2_BENCH_SKIP_SMALL"):
        small = small_corpus(e, np, with_cpu=not args.no_cpu_baseline)
    if rank == 0:
        k_ms = kernel_ms_total / kernel_passes
        frames_per_launch = S
        # PMC traffic cannot be collected inside this process: it comes from separate rocprofv3 --pmc passes over this
        # same command (tools/summarize_profiles.py -> profiles/r05_traffic*.json).  The file records the hash of the
        # kernel sources it was measured on; a figure measured on other sources is reported as stale (null).
        traffic, traffic_detail = None, None
        tname = ("r05_traffic" if prefiltered else "r05np_traffic") + ("" if M == 1024 else f"_M{M}") + ("" if prefiltered else "_noprefilter") + ".json"
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath) and S == FRAMES_PER_GPU:
            try:
                tj = json.load(open(tpath))
                fresh = tj.get("kernel_sources_sha16") == kernel_sources_sha16()
                traffic = tj.get("hbm_bytes_per_launch") if fresh else None
                traffic_detail = {"fetch_bytes": tj.get("fetch_bytes"), "write_bytes": tj.get("write_bytes"),
                                  "algorithmic_bytes": tj.get("algorithmic_bytes_per_launch"),
                                  "measured_on_sources": tj.get("kernel_sources_sha16"),
                                  "stale": not fresh, "note": tj.get("note")}
            except Exception:
                traffic = None
        alg_tf = FLOP_PER_FRAME_PASS * frames_per_launch / (k_ms * 1e-3) / 1e12
        achieved_gbs = BYTES_PER_FRAME_PASS * frames_per_launch / (k_ms * 1e-3) / 1e9
        roofline_algorithmic = None
        if prefiltered:
            # dominant kernel: the prefiltered sweep.  The work it EXECUTES is 15 f16 MFMA k-steps per (frame, codeword)
            # pair (exact integer limb products) -- priced against the dense f16 MFMA peak of the guide; the FP64 chain
            # runs only for the two certified candidates of a frame.
            # k-steps of v_mfma_f32_32x32x16_f16 executed per (frame, codeword) pair: 15 for a one-stage sweep; two stages: 8 for
            # every pair + all 15 again for the flagged (tile, column block) jobs (their share is measured by the kernel)
            ksteps = (8 + 15 * flagged_frac) if (two_stage and flagged_frac >= 0) else 15.0
            exec_tf = 2 * 16 * ksteps * M * frames_per_launch / (k_ms * 1e-3) / 1e12
            if sweep_kind == 3:
                kname, kdesc = "k_sweep_cand", (
                    "k_sweep_cand<37, two-stage, fused> over frames grouped by cell (one counting sort per level), two blocks of "
                    "64 slots per turn of a wave (a loaded codeword tile serves four coarse jobs): exact f16-limb "
                    "prefilter on v_mfma_f32_32x32x16_f16 in two stages (8 coarse k-steps for every codeword tile, all 15 + top-3 "
                    "keys for the tiles a rigorous bound cannot rule out), FP64 rows gathered into LDS by LDS-DMA, the certified "
                    "top two evaluated as lane-per-frame v_fma_f64 chains, symbols / distortion sums out, the contributions to "
                    "the cell sums reduced in the block and added with one atomic per row element (exact int64); then, for "
                    "uncertified frames, k_pass_mfma<37,2,256,2>")
            elif sweep_kind == 2:
                kname, kdesc = "k_sweep_cand", (
                    "k_sweep_cand<37> (candidate sweep, frames in their natural order) + k_finish (exact evaluation, outputs, "
                    "records) + k_reduce_records")
            else:
                kname, kdesc = "k_pass_pre", (
                    "k_pass_pre_lds<37> (round 4's fused kernel: f16-limb prefilter + top-3 keys, FP64 frames of the block staged "
                    "in LDS, lane-per-frame v_fma_f64 chains, contributions recorded for k_reduce_records or added as a burst)")
            roofline = {
                "bound": "mfma",
                "kernel": kdesc,
                "achieved": exec_tf,
                "peak": F16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": exec_tf / F16_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "kernel_ms": k_ms,
                "accumulate_kernel_ms": (pass_kernels_ms_total - kernel_ms_total) / kernel_passes,
                "launches": kernel_passes,
                "trace_dispatches": {"kernel": kname, "first": launches_before[1] if kname == "k_sweep_cand" else launches_before[0],
                                     "count": timed_sweep if kname == "k_sweep_cand" else timed_lds,
                                     "plain_first": launches_before[2], "plain_count": timed_plain},
                "work_per_launch": f"{2 * 16 * ksteps:.1f} f16 MFMA flop x {M} codewords x {S} frames "
                                   f"(limb products actually issued: {ksteps:.2f} k-steps per pair"
                                   + (f", flagged fraction {flagged_frac:.3f}" if two_stage and flagged_frac >= 0 else "") + ")",
                "ksteps_per_pair": ksteps,
                "two_stage": bool(two_stage), "flagged_fraction": flagged_frac if two_stage else None,
                "one_stage_equivalent": {
                    "what": "the limb products a one-stage sweep issues (15 k-steps per pair: round 4's count) / this kernel's "
                            "time: what the same pass would need on round 4's kernel to be as fast -- a speed-up figure",
                    "tflops": F16_MFMA_FLOP_PER_FRAME_CODEWORD * M * frames_per_launch / (k_ms * 1e-3) / 1e12,
                    "frac_of_peak": F16_MFMA_FLOP_PER_FRAME_CODEWORD * M * frames_per_launch / (k_ms * 1e-3) / 1e12 / F16_PEAK_TFLOPS},
                "executed_dtype": "f16 limbs (exact integers) -> f32 accumulators; candidates in f64",
                "fallback_frames_last_pass": fallback_frames,
                "note": "dense f16 MFMA peak of the guide (2.5 PF); on random operands the pipe sustains 1.1-1.4 PF "
                        "(power-limited clock, tools/probe/pre_sweep.hip)",
            }
            roofline_algorithmic = {
                "what": "the path's algorithmic FP64 flops (SURVEY 8d: 2*M*(P+1) per frame-pass) / kernel time, against "
                        "the FP64 peak: a speed-up figure, NOT a roofline fraction (the kernel does not execute these "
                        "flops; the plain FP64 sweep, --no-prefilter, runs at 0.80 of this peak)",
                "fp64_equivalent_tflops": alg_tf,
                "fp64_peak_tflops": FP64_PEAK_TFLOPS,
                "ratio": alg_tf / FP64_PEAK_TFLOPS,
            }
        else:
            # the plain sweep is FP64-FMA bound (248 flop/B): useful flops 2*M*(P+1) per frame against the 78.6 TF peak
            roofline = {
                "bound": "mfma",
                "kernel": "k_pass_mfma<37,2,512> (sweep on v_mfma_f64_16x16x4_f64 + argmin + exact accumulate)",
                "achieved": alg_tf,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": alg_tf / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "kernel_ms": k_ms,
                "launches": kernel_passes,
                "trace_dispatches": {"kernel": "k_pass_mfma", "first": launches_before[2], "count": timed_plain},
            }
        out = {
            "metric": f"vq_learn_frames_per_sec_M{M}_P36",
            "value": world * S * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"vq learn, the real M={M} level at P={P}: {S} frames per GPU (config 4 shard: 16M frames "
                            f"over 8 GPUs); per repetition the point where the M={M // 2} level ended is restored (device copies "
                            f"and the rebuild of the codeword images: inside the timed region), split, and passes run until "
                            f"(DDprv-DD)/DD < {EPS} ({L} passes: the first seeded with the parents' sums -- in-family frames "
                            f"add once or not at all --, {L - 1} incremental ones); a step = one such pass; eps={EPS} ladder 2..{M // 2} run untimed first",
                "frames_per_gpu": S,
                "codebook_size": M,
                "prediction_order": P,
                "passes_per_level": L,
                "parallelism": f"frames sharded over {world} rank(s); int64 all-reduce of cell sums per iteration",
                "collective": collective,
                "sweep": "prefiltered (exact f16-limb prefilter + FP64 verification; bit-identical to the plain sweep)"
                         if prefiltered else "plain FP64 MFMA sweep",
                "ladder_seconds_untimed": round(t_ladder, 3),
                "final_avg_distortion": st.avg_distortion,
                "parity": parity,
                "timed_sweep_launches": {"k_sweep_cand": timed_sweep, "k_pass_pre_lds": timed_lds, "plain": timed_plain},
                "steady_state": None if not extras else {
                    "what": "back-to-back iterations on the converged codebook (incremental accumulate nearly idle): "
                            "round 1's headline regime, informational",
                    "ms_per_step": steady_ms,
                    "kernel_ms": steady_kernel_ms,
                    "frames_per_sec": world * S / (steady_ms * 1e-3),
                },
                "learn_end_to_end": None if not extras else {
                    "what": f"whole LBG ladder M=2..{M}, eps=0.05, resident frames, all ranks, one library call",
                    "seconds": round(e2e_s, 4),
                    "frames_per_sec": world * S / e2e_s,
                    "passes_per_level": [lv.passes for lv in e2e_levels],
                    "levels": level_detail,
                    "levels_note": "the same ladder run level by level (one synchronisation per level): sweep-kernel time "
                                   "per pass from HIP events, step = wall time of the level / its passes; bound = what the "
                                   "level's sweep is priced against (HBM for M <= 32, the FP64 matrix pipe for the plain "
                                   "sweep of 64 <= M <= 128, the f16 limb products actually issued for the prefiltered "
                                   "levels; the first, full pass of M = 256 runs the plain sweep)",
                },
                "quantize_frames_per_sec_device_resident": q_rate,
                "strong_scaling_16M": big,
                "small_corpus": small,
            },
            "roofline": roofline,
            "roofline_algorithmic": roofline_algorithmic,
            "roofline_hbm": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "note": "algorithmic 306 B per frame-pass; at M=1024 the sweep, not HBM, bounds the pass (SURVEY 8d)",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(e, np)
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    sess.close()
    if parity is not None and not parity["ok"]:
        raise SystemExit(f"bench.py: PARITY FAILED on rank {rank}: {parity}")


if __name__ == "__main__":
    main()
