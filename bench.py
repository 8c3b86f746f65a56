#!/usr/bin/env python3
"""bench.py -- VQ-learn frames/s at M=1024, P=36 on N MI355X (BASELINE.json metric).

A "step" is one full LBG iteration at M=1024 over the resident shard of every rank:
  sweep+accumulate kernel (K1+K2)  ->  int64 all-reduce of the cell sums (RCCL, N>1)
  ->  level statistics (the host reads DD for the convergence test)  ->  centroid update (K3/K4).
Frames are synthetic (seeded, counter based: rank r holds frames [r*S, (r+1)*S) of one stream) and
resident in HBM before the timed region.  Weak scaling: S = 2^21 frames per GPU (config 4's shard).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
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
        "kind": "port",
        "sample": f"{reps} assignment passes over {n} synthetic frames at M={M}, P={P} "
                  f"({secs:.1f} s; oracle source built with the reference's flags -O3 -ffast-math -fopenmp, {variant})",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-gpu", type=int, default=FRAMES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefilter", action="store_true",
                    help="keep every pass on the plain FP64 sweep (A/B against the prefiltered sweep; same results)")
    ap.add_argument("--backend", default="nccl", help="process-group backend for N > 1 (nccl = RCCL over xGMI; "
                    "gloo lets several ranks share one GPU when rehearsing the N > 1 path)")
    args = ap.parse_args()

    if args.no_prefilter:
        os.environ["ECOZ2_VQ_PREFILTER"] = "0"
    import numpy as np
    import torch
    import torch.distributed as dist

    import ecoz2rs_amd as e
    from ecoz2rs_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(args.backend)

    S = args.frames_per_gpu
    lo = rank * S
    frames = e.synth.synth_frames(SEED, N_CLASSES, P, lo, S)

    sess = e.VqSession(P, device=local)
    parallel.bind_torch_stream(sess, local)  # session kernels + RCCL collectives on one torch stream
    if world > 1:
        sess.set_allreduce(parallel.make_allreduce(local), rank, world)
    sess.set_frames(frames)  # H2D + blocked re-layout; resident from here on
    del frames
    sess.prepare()
    sess.init_codebook()
    os.environ["ECOZ2_VQ_QUIET"] = "1"
    t_ladder = time.time()
    levels = sess.learn(0.05, M // 2)  # real LBG ladder 2..512 (untimed) -> realistic codebook state
    torch.cuda.synchronize()
    t_ladder = time.time() - t_ladder
    sess.grow()  # M = 1024
    sym = torch.empty(S, dtype=torch.int16, device=f"cuda:{local}")
    dmin = torch.empty(S, dtype=torch.float64, device=f"cuda:{local}")

    def step():  # sweep + accumulate (+ all-reduce) -> level statistics -> centroid update
        return sess.iterate(sym, dmin)

    for _ in range(args.warmup):
        step()
    sess.enable_timing(True)  # HIP events around the sweep kernel of every pass, summed inside the library

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms_total, kernel_passes = sess.timing_total()
    assert kernel_passes == args.steps
    prefiltered, fallback_frames = sess.last_pass_info()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- secondary figures (SURVEY 8d ii / iii), outside the timed region, informational --------------------
    sess.enable_timing(False)
    sess.init_codebook()
    fence()
    t0 = time.perf_counter()
    e2e_levels = sess.learn(0.05, M)  # the whole ladder 2..1024 with the real convergence rule
    fence()
    e2e_s = time.perf_counter() - t0
    q_rate = None
    if world == 1:
        fr = torch.from_numpy(e.synth.synth_frames(SEED, N_CLASSES, P, lo, S)).cuda()
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sess.quantize_device(fr, S, sym, dmin)
            sess.synchronize()
            q_rate = max(q_rate or 0.0, S / (time.perf_counter() - t0))
        del fr

    if rank == 0:
        k_ms = kernel_ms_total / kernel_passes
        frames_per_launch = S
        # PMC traffic cannot be collected inside this process: it comes from the separate rocprofv3 --pmc passes
        # over this same command, recorded in profiles/traffic.json (FETCH_SIZE doubled per the gfx950 note)
        traffic, traffic_detail = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_prefilter.json" if prefiltered else "traffic.json")
        if os.path.exists(tpath) and S == FRAMES_PER_GPU:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_detail = {"fetch_bytes": tj.get("fetch_bytes"), "write_bytes": tj.get("write_bytes"),
                                  "algorithmic_bytes": tj.get("algorithmic_bytes_per_launch"),
                                  "note": tj.get("note")}
            except Exception:
                traffic = None
        achieved_tf = FLOP_PER_FRAME_PASS * frames_per_launch / (k_ms * 1e-3) / 1e12
        achieved_gbs = BYTES_PER_FRAME_PASS * frames_per_launch / (k_ms * 1e-3) / 1e9
        if prefiltered:
            # dominant kernel: the prefiltered sweep.  Its matrix work is 15 f16 MFMA k-steps per (frame, codeword)
            # (exact integer limb products); the FP64 chain runs only for the two certified candidates of a frame.
            exec_tf = F16_MFMA_FLOP_PER_FRAME_CODEWORD * M * frames_per_launch / (k_ms * 1e-3) / 1e12
            kernel_name = ("k_pass_pre<37,2,512> (exact f16-limb prefilter on v_mfma_f32_32x32x16_f16 + top-3 keys, "
                           "FP64 evaluation of the certified top two on v_mfma_f64_16x16x4_f64, incremental exact "
                           "accumulate); uncertified frames: k_pass_mfma<37,2,512,2>")
            # by the contract: ALGORITHMIC flops of the path (2*M*(P+1) FP64 flop per frame-pass) per kernel
            # time against the FP64 peak.  The ratio exceeds 1 because the kernel does not execute most of those flops:
            # it proves, per frame, which two codewords can win and runs the FP64 chain for those only.
            roofline = {
                "bound": "mfma",
                "kernel": kernel_name,
                "achieved": achieved_tf,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tf / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "kernel_ms": k_ms,
                "note": "algorithmic FP64 flops / kernel time; > 1 is the algorithmic gain of the exact prefilter, not a "
                        "measurement artefact (results are bit-identical to the plain FP64 sweep, which runs at 0.80 of "
                        "this peak: --no-prefilter).  The work actually issued is priced in roofline_executed.",
                "fallback_frames_last_pass": fallback_frames,
            }
            roofline_executed = {
                "bound": "mfma",
                "what": "limb products actually issued: 15 k-steps of v_mfma_f32_32x32x16_f16 per (frame, codeword) pair",
                "achieved": exec_tf,
                "peak": F16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": exec_tf / F16_PEAK_TFLOPS,
                "executed_dtype": "f16 limbs (exact integers) -> f32 accumulators; candidates in f64",
                "note": "dense f16 MFMA peak of the guide (2.5 PF); on random operands the pipe sustains 1.1-1.25 PF "
                        "(power), the sweep phase alone runs at 1.13 PF (tools/probe/pre_sweep.hip)",
            }
        else:
            # the plain sweep is FP64-FMA bound (248 flop/B): useful flops 2*M*(P+1) per frame against the 78.6 TF peak
            roofline = {
                "bound": "mfma",
                "kernel": "k_pass_mfma<37,2,512> (sweep on v_mfma_f64_16x16x4_f64 + argmin + exact accumulate)",
                "achieved": achieved_tf,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tf / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_detail": traffic_detail,
                "kernel_ms": k_ms,
            }
            roofline_executed = None
        out = {
            "metric": "vq_learn_frames_per_sec_M1024_P36",
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
                "workload": f"vq learn LBG iteration at M={M}, P={P}: {S} frames per GPU "
                            f"(config 4 shard: 16M frames over 8 GPUs), eps=0.05 ladder 2..{M // 2} run untimed first",
                "frames_per_gpu": S,
                "codebook_size": M,
                "prediction_order": P,
                "parallelism": f"frames sharded over {world} rank(s); int64 all-reduce of cell sums per iteration",
                "sweep": "prefiltered (exact f16-limb prefilter + FP64 verification; bit-identical to the plain sweep)"
                         if prefiltered else "plain FP64 MFMA sweep",
                "ladder_seconds_untimed": round(t_ladder, 3),
                "final_avg_distortion": st.avg_distortion,
                "learn_end_to_end": {
                    "what": f"whole LBG ladder M=2..{M}, eps=0.05, resident frames, all ranks",
                    "seconds": round(e2e_s, 4),
                    "frames_per_sec": world * S / e2e_s,
                    "passes_per_level": [lv.passes for lv in e2e_levels],
                },
                "quantize_frames_per_sec_device_resident": q_rate,
            },
            "roofline": roofline,
            "roofline_executed": roofline_executed,
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
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
