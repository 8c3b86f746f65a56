#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of the prefiltered bench (gpurun_out/prof_pf_*) into the summaries under profiles/.
The timed steps of `bench.py --steps K --warmup W` are k_pass_pre dispatches [6 + W, 6 + W + K): 6 belong to the
untimed ladder (M = 256, 512), the last 9 to the end-to-end ladder.
usage: tools/summarize_prefilter_profiles.py [gpurun_out] [profiles]"""
import csv, glob, json, os, shutil, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles"
T, M = 1 << 21, 1024


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1]


def pre_rows(path):
    return [r for r in csv.DictReader(open(path)) if "k_pass_pre" in r["Kernel_Name"]]


def ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def counters(path, W, K):
    per, order = {}, []
    for r in pre_rows(path):
        d = r["Dispatch_Id"]
        if d not in per:
            per[d] = {}
            order.append(d)
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    sel = [per[d] for d in order][6 + W:6 + W + K]
    return {k: sum(x[k] for x in sel) / len(sel) for k in sel[0]}, sel


kt = newest(f"{SRC}/prof_pf_kt/*/*_kernel_trace.csv")
sel = pre_rows(kt)[9:29]
d = [ms(r) for r in sel]
fb = [r for r in csv.DictReader(open(kt)) if "k_pass_mfma<37, 2, 256, 2>" in r["Kernel_Name"]][9:29]
out = {"kernel": "e2vq::k_pass_pre<37, 2, 512>", "dispatches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d),
       "max_ms": max(d), "vgpr": sel[0].get("VGPR_Count"), "accum_vgpr": sel[0].get("Accum_VGPR_Count"),
       "lds_bytes": sel[0].get("LDS_Block_Size"), "fallback_kernel": "e2vq::k_pass_mfma<37, 2, 256, 2>",
       "fallback_kernel_avg_ms": sum(ms(r) for r in fb) / len(fb),
       "source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20; "
                 "the timed steps are k_pass_pre dispatches 9..28"}
json.dump(out, open(f"{DST}/r01_prefilter_pass_kernel.json", "w"), indent=1)
shutil.copy(newest(f"{SRC}/prof_pf_kt/*/*_kernel_stats.csv"), f"{DST}/r01_prefilter_kernel_stats.csv")
print(json.dumps(out, indent=1))

f, fsel = counters(newest(f"{SRC}/prof_pf_fetch/*/*_counter_collection.csv"), 2, 6)
w, wsel = counters(newest(f"{SRC}/prof_pf_write/*/*_counter_collection.csv"), 2, 6)
tj = {"kernel": "k_pass_pre<37,2,512> at M=1024, 2^21 frames per launch (bench.py --steps 6 --warmup 2: the 6 timed "
                "steps, incremental accumulate active)",
      "FETCH_SIZE_KB_raw": f["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w["WRITE_SIZE"],
      "FETCH_SIZE_KB_per_step": [x["FETCH_SIZE"] for x in fsel], "WRITE_SIZE_KB_per_step": [x["WRITE_SIZE"] for x in wsel],
      "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled "
                    "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for atomics / 16-B stores",
      "fetch_bytes": f["FETCH_SIZE"] * 2048, "write_bytes": w["WRITE_SIZE"] * 1024,
      "hbm_bytes_per_launch": f["FETCH_SIZE"] * 2048 + w["WRITE_SIZE"] * 1024, "algorithmic_bytes_per_launch": 306 * T,
      "note": "reads: 224 B f16 limb image + 4 B tolerance + 296 B FP64 frame + 2 B previous cell per frame (1.10 GB) and "
              "whatever part of the codeword tile images (16.8 GB of L2 reads per launch) misses L2; writes: 2 B symbol + "
              "8 B distortion + 2 B cell per frame (25 MB) and the incremental accumulate: 4 distortion elements per frame "
              "(67 MB) + 2 x 75 int64 elements per frame that changed cell, falling from step to step"}
json.dump(tj, open(f"{DST}/traffic_prefilter.json", "w"), indent=1)

c1, _ = counters(newest(f"{SRC}/prof_pf_sq1/*/*_counter_collection.csv"), 2, 6)
c2, _ = counters(newest(f"{SRC}/prof_pf_sq2/*/*_counter_collection.csv"), 2, 6)
d2 = [ms(r) for r in pre_rows(newest(f"{SRC}/prof_pf_sq2/*/*_kernel_trace.csv"))[8:14]]
d1 = [ms(r) for r in pre_rows(newest(f"{SRC}/prof_pf_sq1/*/*_kernel_trace.csv"))[8:14]]
c = {**c1, **c2}
cyc = c["GRBM_GUI_ACTIVE"] / 8
f16 = (T // 32) * (M // 32) * 15
sq = {"kernel": tj["kernel"].replace("incremental accumulate active", "two PMC-only passes"), "counters": c,
      "kernel_ms_under_pmc": [sum(d1) / len(d1), sum(d2) / len(d2)],
      "expected_f16_mfma_instructions": f16, "fp64_mfma_instructions": c["SQ_INSTS_MFMA"] - f16,
      "mfma_busy_cycles_check": {"f16 x 32 + fp64 x 64": f16 * 32 + (c["SQ_INSTS_MFMA"] - f16) * 64,
                                 "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"]},
      "kernel_cycles_under_pmc": cyc, "clock_GHz_under_pmc": cyc / (sum(d2) / len(d2) * 1e-3) / 1e9,
      "mfma_pipe_busy_fraction": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
      "other_valu_instructions_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
      "note": "SQ_INSTS_MFMA = f16 prefilter MFMAs (blocks32 x tiles32 x 15) + FP64 evaluation MFMAs (9 per 16 frames and "
              "candidate); busy cycles summed over SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs"}
json.dump(sq, open(f"{DST}/r01_prefilter_sq_counters.json", "w"), indent=1)
print({k: sq[k] for k in ("kernel_ms_under_pmc", "clock_GHz_under_pmc", "mfma_pipe_busy_fraction",
                          "other_valu_instructions_per_mfma", "fp64_mfma_instructions")})
print("fetch raw KB", f["FETCH_SIZE"], "write raw KB per step", [round(x["WRITE_SIZE"]) for x in wsel])
