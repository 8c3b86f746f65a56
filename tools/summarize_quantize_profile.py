#!/usr/bin/env python3
"""rocprofv3 outputs of tools/quantize_profile.py (gpurun_out/<tag>_{kt,fetch,write,sq1}) -> profiles/<tag>_{pass_kernel,traffic,
sq_counters}.json for the fused quantize kernel k_pass_pre<37, 6, 512> (config 3).   usage: summarize_quantize_profile.py <tag>"""
import csv
import glob
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r06q"
SRC = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
DST = sys.argv[3] if len(sys.argv) > 3 else "profiles"


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1]


def line(name):
    for ln in open(f"{SRC}/{TAG}_{name}.json"):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"no JSON line in {SRC}/{TAG}_{name}.json")


def is_q(name):  # the fused quantize kernel (mangled: _Float16 vector parameters defeat the demangler), not round 4's k_pass_pre_lds
    return "k_pass_pre" in name and "k_pass_pre_lds" not in name


def ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def rows(path, reps):
    return [r for r in csv.DictReader(open(path)) if is_q(r["Kernel_Name"])][-reps:]  # (the last REPS: the timed calls)


def counters(path, reps):
    per, order = {}, []
    for r in csv.DictReader(open(path)):
        if not is_q(r["Kernel_Name"]):
            continue
        d = r["Dispatch_Id"]
        if d not in per:
            per[d] = {}
            order.append(d)
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    sel = [per[d] for d in order][-reps:]
    return {k: sum(x[k] for x in sel) / len(sel) for k in sel[0]}


b = line("kt")
R, T, M = b["reps"], b["frames"], b["codebook_size"]
kt = rows(newest(f"{SRC}/{TAG}_kt/*/*_kernel_trace.csv"), R)
d = [ms(r) for r in kt]
fb = [r for r in csv.DictReader(open(newest(f"{SRC}/{TAG}_kt/*/*_kernel_trace.csv"))) if "k_pass_mfma" in r["Kernel_Name"]][-R:]
out = {"kernel": kt[0]["Kernel_Name"][:60] + "...  (k_pass_pre<37, 6, 512>: fused quantize)", "dispatches": len(d), "avg_ms": sum(d) / len(d),
       "min_ms": min(d), "max_ms": max(d), "fallback_sweep_avg_ms": sum(ms(r) for r in fb) / max(1, len(fb)),
       "frames_per_launch": T, "codebook_size": M, "event_ms_of_the_same_calls": b["avg_ms"],
       "frames_per_sec_by_events": b["frames_per_sec"],
       "f16_mfma_tflops_executed": 2 * 16 * 15.0 * M * T / (sum(d) / len(d) * 1e-3) / 1e12,
       "frac_of_2500_TF": 2 * 16 * 15.0 * M * T / (sum(d) / len(d) * 1e-3) / 1e12 / 2500.0,
       "lds_bytes": kt[0].get("LDS_Block_Size"), "kernel_sources_sha16": b["kernel_sources_sha16"],
       "source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/quantize_profile.py"}
json.dump(out, open(f"{DST}/{TAG}_pass_kernel.json", "w"), indent=1)
print(json.dumps(out, indent=1))
try:
    f = counters(newest(f"{SRC}/{TAG}_fetch/*/*_counter_collection.csv"), R)
    w = counters(newest(f"{SRC}/{TAG}_write/*/*_counter_collection.csv"), R)
    tj = {"kernel": out["kernel"], "frames_per_launch": T, "FETCH_SIZE_KB_raw": f["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w["WRITE_SIZE"],
          "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM "
                        "section); WRITE_SIZE exact", "fetch_bytes": f["FETCH_SIZE"] * 2048, "write_bytes": w["WRITE_SIZE"] * 1024,
          "hbm_bytes_per_launch": f["FETCH_SIZE"] * 2048 + w["WRITE_SIZE"] * 1024, "algorithmic_bytes_per_launch": 298 * T,
          "ratio_to_algorithmic": (f["FETCH_SIZE"] * 2048 + w["WRITE_SIZE"] * 1024) / (298 * T),
          "kernel_sources_sha16": b["kernel_sources_sha16"]}
    json.dump(tj, open(f"{DST}/{TAG}_traffic.json", "w"), indent=1)
    print(json.dumps(tj, indent=1))
except (IndexError, FileNotFoundError, KeyError) as ex:
    print("no PMC traffic passes:", ex)
try:
    c = counters(newest(f"{SRC}/{TAG}_sq1/*/*_counter_collection.csv"), R)
    c2 = counters(newest(f"{SRC}/{TAG}_sq2/*/*_counter_collection.csv"), R)
    d2 = [ms(r) for r in rows(newest(f"{SRC}/{TAG}_sq2/*/*_kernel_trace.csv"), R)]
    c = {**c, **c2}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    sq = {"kernel": out["kernel"], "counters": c, "kernel_cycles_under_pmc": cyc, "clock_GHz_under_pmc": cyc / (sum(d2) / len(d2) * 1e-3) / 1e9,
          "mfma_pipe_busy_fraction": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
          "other_valu_instructions_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
          "expected_f16_mfma_instructions": (T // 32) * (M // 32) * 15, "kernel_sources_sha16": b["kernel_sources_sha16"]}
    json.dump(sq, open(f"{DST}/{TAG}_sq_counters.json", "w"), indent=1)
    print({k: sq[k] for k in ("clock_GHz_under_pmc", "mfma_pipe_busy_fraction", "other_valu_instructions_per_mfma")})
except (IndexError, FileNotFoundError, KeyError) as ex:
    print("no SQ passes:", ex)
