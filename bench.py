#!/usr/bin/env python3
"""bench.py -- VQ-learn frames/s at M=1024, P=36 on N MI355X (BASELINE.json metric, config 4).

A "step" is one LBG pass of the REAL M=1024 level over the resident shard of every rank:
  sweep+accumulate kernel (K1+K2)  ->  int64 all-reduce of the cell sums (RCCL, N>1)
  ->  level statistics (the host reads DD for the convergence test)  ->  centroid update (K3/K4).
The timed region repeats the level as the ladder runs it: restore the point where the M=512 level ended (device copies
of the codebook, its rows and cells, and the rebuild of the codeword images: inside the timed region), split to M=1024,
then passes until (DDprv-DD)/DD < eps ends the level (3 passes on this data: the first seeded with the parents' sums,
two incremental ones), through the library's own e2vq_learn.  K steps = K such passes (whole
levels; a remainder of K is run as the leading passes of one more level).
Frames are synthetic (seeded, counter based: rank r holds frames [r*S, (r+1)*S) of one stream) and resident in HBM before
the timed region.  The workload is BASELINE config 4: 2^24 frames sharded over the N GPUs -- S = 2^24 / N per GPU, so N = 1
holds the whole set (18 GB resident) and N = 8 the 2^21-frame shards: "scaling": "strong".  (--frames-per-gpu S fixes the
per-GPU shard instead: weak scaling.)  At N = 1 the line also carries `config.weak_scaling_anchor`: the same timed region on
one 2^21-frame shard -- what each of eight GPUs does between two exchanges --, the per-level ladder table on that shard,
`config.quantize` (config 3's kernel with its own roofline) and `config.robustness` (the level on other data shapes).

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
TOTAL_FRAMES = 1 << 24      # BASELINE.json config 4: 16 M frames, sharded over the GPUs of one node (all of them on one GPU at N = 1)
ANCHOR_FRAMES = 1 << 21     # ... its per-GPU shard at N = 8
FRAMES_PER_GPU = ANCHOR_FRAMES
PROFILE_ROUND = "r06"       # profiles/<tag>_traffic*.json quoted in roofline.traffic (tools/summarize_profiles.py)
PROFILE_TAGS = {1024: "r06", 512: "r06m512", 256: "r06m256"}
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
    # (2^20 frames = 310 MB: the sample streams from memory like the real pass, not from the caches)
    frames = e.synth.synth_frames(SEED, N_CLASSES, P, 0, 1 << 20)
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
    out = {"what": f"notes.md:122-153: {T} training vectors, eps 0.05, M = 2 ... {MM}, P = {P}; synthetic frames SHAPED like that "
                   f"corpus (e2vq_synth_frames_kind 1: no classes, reflections on a smooth trajectory around the Levinson recursion of "
                   f"the vector notes.md:80-85 prints, r[0] = 1/E about 2-3; seed {SEED}) -- rounds 4 and 5 ran these sizes on the "
                   f"20-class generator, whose ladder is kept as resident_ladder_seconds_class_generator",
           "frames": T, "max_codebook_size": MM}
    try:
        frames = e.synth.synth_frames_kind(SEED, 1, 6, 0.01, P, 0, T)
        out["mean_r0"] = float(frames[:, 0].mean())
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
            out["resident_ladder_passes_per_level"] = [x.passes for x in whole]
            out["resident_ladder_us_per_pass"] = round(1e6 * (time.perf_counter() - t0) / max(1, out["resident_ladder_passes"]), 1)
            out["final_avg_distortion"] = whole[-1].avg_distortion  # (notes.md:150-153: 0.0587 at M = 2048 on the real corpus)
            out["decided"] = dict(zip(("one_stage_until_M", "plain_sweep_from_M", "uncertified_frames_last_pass"), s.sweep_policy_state()))
        with e.VqSession(P, device=0) as s:  # the same sizes on the 20-class generator (what rounds 4 and 5 reported)
            s.set_frames(e.synth.synth_frames(SEED, N_CLASSES, P, 0, T))
            s.prepare()
            for rep in range(2):
                s.init_codebook()
                s.synchronize()
                t0 = time.perf_counter()
                wc = s.learn(0.05, MM)
                s.synchronize()
                out["resident_ladder_seconds_class_generator"] = round(time.perf_counter() - t0, 5)
                out["resident_ladder_passes_class_generator"] = sum(x.passes for x in wc)
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
    ap.add_argument("--frames-per-gpu", type=int, default=None,
                    help="default: BASELINE config 4's 2^24 frames divided by --gpus (strong scaling: the whole set on one GPU "
                         "at N = 1, 2^21 per GPU at N = 8); given explicitly, that many frames on every GPU (weak scaling)")
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


ROBUSTNESS_GENERATORS = [
    # (name, kind, n_classes, noise, what)
    ("bench_20_classes", 0, N_CLASSES, 0.05, "the bench data: 20 class prototypes, within-class noise 0.05"),
    ("one_class", 0, 1, 0.05, "ONE prototype: no between-class structure at all"),
    ("200_classes", 0, 200, 0.05, "200 class prototypes, within-class noise 0.05"),
    ("20_classes_4x_noise", 0, N_CLASSES, 0.20, "20 class prototypes with four times the within-class noise"),
    ("continuum_r0_2to3", 1, 6, 0.01,
     "no classes: reflections on a smooth trajectory around the mean of the reference's whale-song predictor file "
     "(notes.md:80-85: r[0] = 1/E about 2-3; built as lpc_rs.rs:116-131 builds its vectors: r / prediction error)"),
]
R04_KERNEL_MS_2P21 = 1.04  # round 4's kernel on 2^21 frames at M = 1024: no generator may be slower than that


class LevelBench:
    """one session on S resident frames with the ladder run up to M / 2 and that point saved: times the real level M"""

    def __init__(self, e, torch, comm, local, frames, M, eps=0.05):
        self.e, self.torch, self.comm, self.local, self.M, self.EPS = e, torch, comm, local, M, eps
        self.S = frames.shape[0]
        self.sess = e.VqSession(P, device=local)
        comm.attach(self.sess)
        self.sess.set_frames(frames)  # H2D + re-layout; resident from here on
        self.sess.prepare()
        self.sess.init_codebook()
        t0 = time.time()
        self.levels_below = self.sess.learn(eps, M // 2)  # real LBG ladder 2..M/2 (untimed) -> realistic codebook state
        self.sess.synchronize()
        self.ladder_below_s = time.time() - t0
        self.sess.save_state()  # the converged M / 2 codebook, its DD, and the rows and cells the seeded first pass starts from

    def whole_level(self):
        """the level through the library's LBG driver (e2vq_learn): restore, split, passes until convergence"""
        self.sess.restore_state()
        return self.sess.learn(self.EPS, self.M)[0]

    def leading_passes(self, n):
        """the first n passes of the level, step by step (same calls, same order as e2vq_learn's loop)"""
        sess = self.sess
        sess.restore_state()
        sess.grow()
        for i in range(n):
            sess.run_pass()
            st_ = sess.pass_stats()
            if i + 1 < n:
                sess.update()
        return st_

    def fence(self):
        self.sess.synchronize()
        self.torch.cuda.synchronize(self.local)
        self.comm.barrier()
        self.sess.synchronize()

    def timed(self, steps, warmup):
        """W warm-up passes (whole levels), then EXACTLY `steps` passes between two fences; the wall time is this rank's"""
        sess = self.sess
        lv = self.whole_level()
        L = lv.passes
        done = L
        while done < warmup:
            self.whole_level()
            done += L
        sess.enable_timing(True)  # HIP events around the sweep kernel of every pass, summed inside the library
        sess.enable_collective_timing(True)  # ... and around every call of the exchange
        sess.sweep_executed(reset=True)
        self.fence()
        launches_before = sess.launch_counts_by_kernel()  # (k_pass_pre_lds, k_sweep_cand, plain)
        ar_before = self.comm.counters()
        t0 = time.perf_counter()
        left = steps
        while left >= L:
            st = self.whole_level()
            assert st.passes == L
            left -= L
        if left:
            st = self.leading_passes(left)
        self.fence()
        dt = time.perf_counter() - t0
        ar_now = self.comm.counters()
        r = {"dt": dt, "L": L, "st": st, "steps": steps,
             "ar_timed": {k: ar_now[k] - ar_before[k] for k in ("n", "bytes")}}
        # the dominant kernel alone (the sweep), and the pass's kernels together (sweep + k_reduce_records where the accumulate
        # is a kernel of its own)
        r["kernel_ms_total"], r["kernel_passes"] = sess.timing_sweep_total()
        r["pass_kernels_ms_total"] = sess.timing_total()[0]
        assert r["kernel_passes"] == steps, (r["kernel_passes"], steps)
        r["ar_ms"], r["ar_n"], r["ar_b"] = sess.collective_timing()
        sess.enable_collective_timing(False)
        r["prefiltered"], r["fallback_frames"] = sess.last_pass_info()
        la = sess.launch_counts_by_kernel()
        r["launches_before"] = launches_before
        r["timed_lds"], r["timed_sweep"], r["timed_plain"] = (la[k] - launches_before[k] for k in range(3))
        r["sweep_kind"], r["two_stage"], r["flagged_frac_first_pass"] = sess.last_pass_sweep()
        # what the fused sorted passes of the timed region executed, counted by the kernels: (tile, column block) jobs
        fl, jobs, one = sess.sweep_executed(reset=True)
        r["executed"] = {"flagged_jobs": fl, "two_stage_jobs": jobs, "one_stage_jobs": one}
        r["ksteps"] = (8.0 * jobs + 15.0 * fl + 15.0 * one) / (jobs + one) if (jobs + one) > 0 else 15.0
        r["flagged_frac"] = fl / jobs if jobs > 0 else -1.0
        return r

    def ladder_report(self, prefiltered, collective):
        """the whole ladder 2 .. M in one library call, then level by level (a synchronisation per level: ~30 us each) with the
        sweep kernel's event time: per level {passes, kernel ms per pass, step ms per pass, what bounds the level's sweep,
        fraction of that bound}"""
        sess, S, M = self.sess, self.S, self.M
        sess.enable_timing(False)
        sess.init_codebook()
        self.fence()
        t0 = time.perf_counter()
        e2e_levels = sess.learn(self.EPS, M)
        self.fence()
        e2e_s = time.perf_counter() - t0
        detail = []
        sess.init_codebook()
        m = 2
        while m <= M:
            sess.enable_timing(True)
            sess.enable_collective_timing(collective)
            self.fence()
            t0 = time.perf_counter()
            lvm = sess.learn(self.EPS, m)[0]
            self.fence()
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
            detail.append({"M": m, "passes": lvm.passes, "kernel_ms": kms / max(1, kn), "step_ms": wall / lvm.passes * 1e3,
                           "bound": bound, "bound_ms": bound_ms, "frac_of_bound": bound_ms / (kms / max(1, kn)),
                           "sweep_kind": lkind, "flagged_fraction": lfrac if ltwo else None})
            if collective:
                detail[-1]["allreduce_us_per_call"] = 1e3 * lar_ms / max(1, lar_n)
                detail[-1]["allreduce_ms_per_step"] = lar_ms / max(1, lvm.passes)
                detail[-1]["allreduce_bytes"] = m * self.e.lib.e2vq_row_stride(P) * 8
            m *= 2
        sess.enable_timing(False)
        sess.enable_collective_timing(False)
        return {
            "what": f"whole LBG ladder M=2..{M}, eps={self.EPS}, {S} resident frames per rank, all ranks, one library call",
            "seconds": round(e2e_s, 5),
            "frames_per_sec": self.comm.world * S / e2e_s,
            "passes_per_level": [lv.passes for lv in e2e_levels],
            "levels": detail,
            "levels_note": "the same ladder run level by level (one synchronisation per level): sweep-kernel time per pass "
                           "from HIP events, step = wall time of the level / its passes (with N > 1: allreduce_ms_per_step = "
                           "device time of the exchange per pass, inside step_ms); bound = what the level's sweep is priced "
                           "against (HBM for M <= 32, the FP64 matrix pipe for the plain sweep of M = 64, the f16 limb "
                           "products actually issued for the prefiltered levels)",
        }

    def close(self):
        self.sess.close()


def level_pass_by_pass(lb):
    """the level M of a LevelBench step by step (same calls, same order as e2vq_learn's loop), synchronised around every pass:
    per pass the sweep kernels' event time, the wall time of pass + statistics + update, which sweep ran, how many frames it
    left to the FP64 fallback sweep, and what the host has decided by then"""
    sess = lb.sess
    sess.restore_state()
    sess.grow()
    passes = []
    dd_prev = sess.prev_distortion()
    for i in range(64):
        sess.enable_timing(True)
        sess.synchronize()
        t0 = time.perf_counter()
        sess.run_pass()
        st = sess.pass_stats()
        sess.synchronize()
        wall = time.perf_counter() - t0
        ms = sess.timing_total()[0]
        kd, two, ff = sess.last_pass_sweep()
        one_until, plain_from, unc = sess.sweep_policy_state()
        passes.append({"kernel_ms": ms, "step_ms": wall * 1e3, "sweep_kind": kd, "two_stage": two,
                       "uncertified_frames": unc, "flagged_fraction": ff if (i == 0 and two) else None,
                       "decided": {"one_stage_until_M": one_until, "plain_sweep_from_M": plain_from}})
        ratio = (dd_prev - st.DD) / st.DD
        dd_prev = st.DD
        if i > 0 and not ratio >= lb.EPS:
            break
        sess.update()
    sess.enable_timing(False)
    return passes


def robustness_section(e, torch, comm, local, M, S):
    """Untimed extra (VERDICT r05 task 2): the headline level on data of other shapes.  Per generator: the ladder 2 .. M/2, then
    the level M pass by pass five ways -- the product (the host's switches at their defaults), one stage behind the measuring
    pass, two stages whatever the flagged share, round 4's kernel throughout (k_pass_pre_lds + k_reduce_records: ECOZ2_VQ_ACCUMULATE=records), the plain FP64 sweep -- and the
    level through e2vq_learn prefiltered and plain: codebook bytes, pass count and DD must be identical."""
    import hashlib

    out = {"what": f"the M={M} level on {S} frames of other generators (one GPU, untimed extras): does the two-stage sweep's gain "
                   "survive data without the bench data's cluster structure, do the host's switches (one stage above a flagged "
                   "share; plain sweep above an uncertified share) pick the fastest kernel, and what does a pass cost on data "
                   "shaped like the reference's own corpus?  kernel_ms = the sweep kernels of a pass (HIP events), step_ms = wall "
                   "time of pass + statistics + update, synchronised (the FP64 fallback sweep of uncertified frames is in step_ms "
                   "only)",
           "round4_kernel_ms_on_2p21_frames_bench_data": R04_KERNEL_MS_2P21, "frames": S, "generators": []}

    def brief(passes):
        return {"passes": len(passes), "kernel_ms": [round(p["kernel_ms"], 4) for p in passes],
                "step_ms": [round(p["step_ms"], 4) for p in passes],
                "sweeps": [("plain" if p["sweep_kind"] == 0 else f"kind{p['sweep_kind']}" + ("/two-stage" if p["two_stage"] else "/one-stage"))
                           for p in passes],
                "uncertified_fraction": [None if p["uncertified_frames"] < 0 else round(p["uncertified_frames"] / S, 5) for p in passes],
                "step_ms_per_pass": sum(p["step_ms"] for p in passes) / len(passes)}

    for name, kind, ncls, noise, what in ROBUSTNESS_GENERATORS:
        row = {"generator": name, "what": what, "synth": {"kind": kind, "n_classes": ncls, "noise": noise, "seed": SEED}}
        try:
            frames = e.synth.synth_frames_kind(SEED, kind, ncls, noise, P, 0, S)
            row["mean_r0"] = float(frames[:4096, 0].mean())
            lb = LevelBench(e, torch, comm, local, frames, M)
            sess = lb.sess
            lb.whole_level()  # (warm)
            sess.set_sweep_policy(-1.0, -1.0)  # (defaults; forgets what the warm-up decided)
            prod = level_pass_by_pass(lb)
            row["flagged_fraction_first_pass"] = prod[0]["flagged_fraction"]
            row["product"] = brief(prod)
            row["product"]["decided"] = prod[-1]["decided"]
            sess.set_sweep_policy(0.0, 1.0)   # one stage from the second pass on (round 4's kernel), never the plain sweep
            row["one_stage_after_first_pass"] = brief(level_pass_by_pass(lb))
            sess.set_sweep_policy(1.0, 1.0)   # two stages whatever the flagged share
            row["sorted_two_stage"] = brief(level_pass_by_pass(lb))
            sess.set_prefilter(False)
            row["plain_sweep"] = brief(level_pass_by_pass(lb))
            sess.set_prefilter(True)
            sess.set_sweep_policy(0.45, 0.40)
            # the level through e2vq_learn, prefiltered and plain: identical bytes?
            lv_pre = lb.whole_level()
            cb_pre = sess.get_codebook()
            sess.set_prefilter(False)
            lv_plain = lb.whole_level()
            cb_plain = sess.get_codebook()
            sess.set_prefilter(True)
            row["level_passes_learn"] = lv_pre.passes
            row["codebook_sha16"] = hashlib.sha256(cb_pre.tobytes()).hexdigest()[:16]
            row["equals_plain_sweep"] = bool(cb_plain.tobytes() == cb_pre.tobytes() and lv_plain.passes == lv_pre.passes and
                                             lv_plain.DD.hex() == lv_pre.DD.hex())
            lb.close()
            # round 4's kernel on the same frames (the accumulate is chosen when a session is created)
            old = os.environ.get("ECOZ2_VQ_ACCUMULATE")
            os.environ["ECOZ2_VQ_ACCUMULATE"] = "records"
            try:
                lb4 = LevelBench(e, torch, comm, local, frames, M)
                lb4.sess.set_sweep_policy(-1.0, 1.0)
                lb4.whole_level()
                row["round4_kernel"] = brief(level_pass_by_pass(lb4))
                lb4.close()
            finally:
                if old is None:
                    os.environ.pop("ECOZ2_VQ_ACCUMULATE", None)
                else:
                    os.environ["ECOZ2_VQ_ACCUMULATE"] = old
            del frames
            variants = {k: row[k]["step_ms_per_pass"] for k in ("product", "one_stage_after_first_pass", "sorted_two_stage", "plain_sweep", "round4_kernel")}
            row["step_ms_per_pass"] = {k: round(v, 4) for k, v in variants.items()}
            row["fastest"] = min(variants, key=variants.get)
            row["product_over_fastest"] = variants["product"] / min(variants.values())
        except Exception as ex:  # informational: never fails the bench line by itself
            row["error"] = repr(ex)
        out["generators"].append(row)
    out["all_equal_plain_sweep"] = all(g.get("equals_plain_sweep") for g in out["generators"])
    out["worst_product_over_fastest"] = max((g.get("product_over_fastest", 0.0) for g in out["generators"]), default=None)
    return out


def quantize_section(e, torch, local, sess, S, M, sym, dmin):
    """config 3's kernel on S device-resident frames against the session's M-codeword codebook: frames/s, and the roofline of
    its sweep kernel (HIP events on the session's stream around e2vq_quantize_device: codebook image ready, so the events
    bracket the sweep + the FP64 fallback sweep)"""
    fr = torch.from_numpy(e.synth.synth_frames(SEED, N_CLASSES, P, 0, S)).cuda(local)
    st = torch.cuda.current_stream(local)
    best = None
    for _ in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(local)
        t0 = time.perf_counter()
        ev0.record(st)
        sess.quantize_device(fr, S, sym, dmin)
        ev1.record(st)
        sess.synchronize()
        wall = time.perf_counter() - t0
        ms = ev0.elapsed_time(ev1)
        if best is None or ms < best[0]:
            best = (ms, wall)
    del fr
    ms, wall = best
    exec_tf = 2 * 16 * 15.0 * M * S / (ms * 1e-3) / 1e12
    return {
        "frames": S, "codebook_size": M,
        "frames_per_sec_device_resident": S / wall,
        "kernel_ms": ms,
        "roofline_quantize": {
            "bound": "mfma", "kernel": "k_pass_pre<37,6,512> (fused quantize: limb images built from the row-major payload in the "
                                       "kernel, one-stage f16-limb sweep with all 15 k-steps, FP64 evaluation of the certified top "
                                       "two from the frames staged in LDS) + the FP64 fallback sweep of uncertified frames",
            "achieved": exec_tf, "peak": F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": exec_tf / F16_PEAK_TFLOPS,
            "work_per_launch": f"480 f16 MFMA flop x {M} codewords x {S} frames (15 k-steps per pair: frames in their natural "
                               "order flag 93-99 % of the tiles, nothing to skip)",
            "hbm_algorithmic_GBs": 298 * S / (ms * 1e-3) / 1e9, "hbm_frac": 298 * S / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "timing": "HIP events on the stream the kernels run on (the session is bound to torch's current stream)",
        },
    }


def run(args, comm):
    import numpy as np
    import torch

    import ecoz2rs_amd as e

    M = args.codebook_size
    FLOP_PER_FRAME_PASS = 2 * M * (P + 1)
    rank, world, local = comm.rank, comm.world, comm.local
    # config 4 (BASELINE.json): 2^24 frames sharded over the ranks -- one GPU holds them all (18 GB resident)
    strong = args.frames_per_gpu is None
    S = TOTAL_FRAMES // world if strong else args.frames_per_gpu
    lo = rank * S
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    frames = e.synth.synth_frames(SEED, N_CLASSES, P, lo, S)
    lb = LevelBench(e, torch, comm, local, frames, M)
    del frames
    sess, EPS = lb.sess, lb.EPS
    t_ladder = lb.ladder_below_s
    sym = torch.empty(S, dtype=torch.int16, device=f"cuda:{local}")
    dmin = torch.empty(S, dtype=torch.float64, device=f"cuda:{local}")

    tm = lb.timed(args.steps, args.warmup)
    L, st = tm["L"], tm["st"]
    kernel_ms_total, kernel_passes, pass_kernels_ms_total = tm["kernel_ms_total"], tm["kernel_passes"], tm["pass_kernels_ms_total"]
    prefiltered, fallback_frames = tm["prefiltered"], tm["fallback_frames"]
    launches_before, timed_lds, timed_sweep, timed_plain = tm["launches_before"], tm["timed_lds"], tm["timed_sweep"], tm["timed_plain"]
    sweep_kind, two_stage, flagged_frac, ksteps = tm["sweep_kind"], tm["two_stage"], tm["flagged_frac"], tm["ksteps"]
    dt = comm.max_over_ranks(tm["dt"])
    collective = comm.describe(tm["ar_timed"])
    if collective is not None:
        # device time between the events the library records around its calls of the exchange (the collective's kernels and
        # their wait for the other ranks), this rank, timed region
        collective["allreduce_calls_timed"] = tm["ar_n"]
        collective["allreduce_bytes_per_call"] = tm["ar_b"] // max(1, tm["ar_n"])
        collective["allreduce_us_per_call"] = 1e3 * tm["ar_ms"] / max(1, tm["ar_n"])
        collective["allreduce_ms_per_step"] = tm["ar_ms"] / max(1, args.steps)

    parity = None
    if not args.no_parity:
        parity = parity_section(args, comm, sess, lb.whole_level, sym, dmin, S, lo, M)

    extras = not args.no_extras
    steady_ms = steady_kernel_ms = None
    ladder = quant = anchor = robust = None
    if extras:
        # ---- steady state (informational): back-to-back iterations on the converged codebook, where the
        # incremental accumulate has almost nothing left to move -- round 1's headline regime, kept as an extra key
        lb.whole_level()
        for _ in range(3):
            sess.iterate(sym, dmin)
        sess.enable_timing(True)
        lb.fence()
        t0 = time.perf_counter()
        for _ in range(10):
            sess.iterate(sym, dmin)
        lb.fence()
        steady_ms = (time.perf_counter() - t0) / 10 * 1e3
        steady_kernel_ms = sess.timing_total()[0] / 10
        # ---- secondary figures (SURVEY 8d ii / iii), outside the timed region, informational --------------------
        ladder = lb.ladder_report(prefiltered, collective is not None)
        if world == 1:
            lb.whole_level()  # (the level's final codebook)
            quant = quantize_section(e, torch, local, sess, min(S, 10_000_000), M, sym, dmin)
    lb_S = S
    if extras and world == 1 and M == 1024 and strong and not os.environ.get("ECOZ2_BENCH_SKIP_ANCHOR"):
        # ---- config 4's per-GPU shard (2^21 frames) on this one GPU: what each of the 8 ranks does between two exchanges --
        # the anchor a SCALE run at N = 8 compares with (shard-size effect apart from the collective's cost)
        try:
            lb.close()
            lb = None
            del sym, dmin
            torch.cuda.empty_cache()
            fa = e.synth.synth_frames(SEED, N_CLASSES, P, 0, ANCHOR_FRAMES)
            la = LevelBench(e, torch, comm, local, fa, M)
            del fa
            ta = la.timed(args.steps, args.warmup)
            anchor = {
                "what": f"the same timed region on config 4's per-GPU shard ({ANCHOR_FRAMES} frames = 2^24 / 8) resident on this one "
                        "GPU: the N = 8 point of the strong-scaling curve without its exchange",
                "frames": ANCHOR_FRAMES, "value": ANCHOR_FRAMES * args.steps / ta["dt"], "ms_per_step": ta["dt"] / args.steps * 1e3,
                "kernel_ms": ta["kernel_ms_total"] / ta["kernel_passes"], "passes_per_level": ta["L"],
                "ksteps_per_pair": ta["ksteps"], "flagged_fraction": ta["flagged_frac"],
                "learn_end_to_end": la.ladder_report(ta["prefiltered"], False),
            }
            la.close()
            if not os.environ.get("ECOZ2_BENCH_SKIP_ROBUSTNESS"):
                robust = robustness_section(e, torch, comm, local, M, ANCHOR_FRAMES)
        except Exception as ex:  # (informational)
            anchor = {"error": repr(ex)}

    small = None
    if extras and rank == 0 and world == 1 and M == 1024 and not os.environ.get("ECOZ2_BENCH_SKIP_SMALL"):
        small = small_corpus(e, np, with_cpu=not args.no_cpu_baseline)
    if rank == 0:
        k_ms = kernel_ms_total / kernel_passes
        frames_per_launch = S
        # PMC traffic cannot be collected inside this process: it comes from separate rocprofv3 --pmc passes over this
        # same command (tools/profile_bench.sh + tools/summarize_profiles.py -> profiles/<tag>_traffic*.json).  The file records
        # the hash of the kernel sources and the frames per launch it was measured on; a figure measured on other sources or
        # another shard size is reported as stale (null).
        traffic, traffic_detail = None, None
        tag = PROFILE_TAGS.get(M, f"{PROFILE_ROUND}m{M}") if prefiltered else f"{PROFILE_ROUND}np"
        tname = f"{tag}_traffic" + ("" if M == 1024 else f"_M{M}") + ("" if prefiltered else "_noprefilter") + ".json"
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                fresh = tj.get("kernel_sources_sha16") == kernel_sources_sha16() and tj.get("frames_per_launch") == S
                traffic = tj.get("hbm_bytes_per_launch") if fresh else None
                traffic_detail = {"file": "profiles/" + tname, "fetch_bytes": tj.get("fetch_bytes"), "write_bytes": tj.get("write_bytes"),
                                  "algorithmic_bytes": tj.get("algorithmic_bytes_per_launch"),
                                  "measured_on_sources": tj.get("kernel_sources_sha16"), "measured_on_frames": tj.get("frames_per_launch"),
                                  "stale": not fresh, "note": tj.get("note")}
            except Exception:
                traffic = None
        alg_tf = FLOP_PER_FRAME_PASS * frames_per_launch / (k_ms * 1e-3) / 1e12
        achieved_gbs = BYTES_PER_FRAME_PASS * frames_per_launch / (k_ms * 1e-3) / 1e9
        roofline_algorithmic = None
        if prefiltered:
            # dominant kernel: the prefiltered sweep.  The work it EXECUTES is f16 MFMA k-steps (exact integer limb products)
            # -- priced against the dense f16 MFMA peak of the guide; the FP64 chain runs only for the two certified
            # candidates of a frame.  k-steps of v_mfma_f32_32x32x16_f16 per (frame, codeword) pair: 15 for a one-stage sweep;
            # two stages: 8 for every pair + all 15 again for the flagged (tile, column block) jobs -- counted by the kernels of
            # EVERY timed pass (e2vq_sweep_executed), jobs-weighted
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
                                   f"(limb products actually issued: {ksteps:.3f} k-steps per pair, counted by the kernels over "
                                   f"all {kernel_passes} timed passes"
                                   + (f"; flagged share of the jobs {flagged_frac:.4f}" if flagged_frac >= 0 else "") + ")",
                "ksteps_per_pair": ksteps,
                "executed_jobs": tm["executed"],
                "two_stage": bool(two_stage), "flagged_fraction": flagged_frac if flagged_frac >= 0 else None,
                "flagged_fraction_first_pass_of_level": tm["flagged_frac_first_pass"] if two_stage else None,
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
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"vq learn, BASELINE config 4: {TOTAL_FRAMES} frames at P={P} sharded over {world} GPU(s) "
                             f"({S} per GPU, all resident)" if strong else f"vq learn, {S} frames per GPU at P={P}") +
                            f", the real M={M} level; per repetition the point where the M={M // 2} level ended is restored (device copies "
                            f"and the rebuild of the codeword images: inside the timed region), split, and passes run until "
                            f"(DDprv-DD)/DD < {EPS} ({L} passes: the first seeded with the parents' sums -- in-family frames "
                            f"add once or not at all --, {L - 1} incremental ones); a step = one such pass; eps={EPS} ladder 2..{M // 2} run untimed first",
                "frames_per_gpu": S,
                "frames_total": world * S,
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
                "steady_state": None if steady_ms is None else {
                    "what": "back-to-back iterations on the converged codebook (incremental accumulate nearly idle): "
                            "round 1's headline regime, informational",
                    "ms_per_step": steady_ms,
                    "kernel_ms": steady_kernel_ms,
                    "frames_per_sec": world * S / (steady_ms * 1e-3),
                },
                "learn_end_to_end": ladder,
                "quantize": quant,
                "quantize_frames_per_sec_device_resident": None if quant is None else quant["frames_per_sec_device_resident"],
                "weak_scaling_anchor": anchor,
                "robustness": robust,
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
    if lb is not None:
        lb.close()
    if parity is not None and not parity["ok"]:
        raise SystemExit(f"bench.py: PARITY FAILED on rank {rank}: {parity}")


if __name__ == "__main__":
    main()
