#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/<tag>_*) into the summaries under profiles/.

The timed dispatches of the dominant kernel are cut out of each whole-run trace with the `trace_dispatches` hint
bench.py prints in its JSON line (first index and count among the launches of that kernel family).
usage: tools/summarize_profiles.py <tag> [gpurun_out] [profiles]"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
DST = sys.argv[3] if len(sys.argv) > 3 else "profiles"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, M = 1 << 21, 1024


def kernel_sources_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ecoz2rs_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1]


def bench_line(name):
    for line in open(f"{SRC}/{TAG}_{name}.json"):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no bench line in {SRC}/{TAG}_{name}.json")


def is_dominant(kname, family):
    # the training sweep itself: k_pass_pre<...> or k_pass_mfma<37, m, 512> (the FP64 fallback sweep is <.., 256, 2>)
    # (rocprofv3 leaves k_pass_pre mangled -- its _Float16 vector parameters defeat the demangler: _ZN4e2vq10k_pass_preILi37E..)
    if family == "k_sweep_cand":  # round 5 (mangled like k_pass_pre)
        return "k_sweep_cand" in kname
    if family == "k_pass_pre":
        return "k_pass_pre" in kname
    head = kname.split("(")[0]
    # (k_pass_small serves M <= 16: a plain training sweep too, counted by the library among the plain launches)
    return ("k_pass_mfma<" in head and not head.rstrip().endswith(", 2>")) or "k_pass_small" in head or "k_pass_generic" in head


def ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def timed_rows(path, hint):
    rows = [r for r in csv.DictReader(open(path)) if is_dominant(r["Kernel_Name"], hint["kernel"])]
    return rows[hint["first"]:hint["first"] + hint["count"]]


def reduce_rows(path, hint):
    """k_reduce_records launches of the timed region: one behind every training sweep of the k_pass_pre family when the
    accumulate is recorded (the default) -- same indices; none otherwise."""
    allr = list(csv.DictReader(open(path)))
    red = [r for r in allr if "k_reduce_records" in r["Kernel_Name"]]
    # (quantize sweeps are of the k_pass_pre family too, but follow the timed region; the fused pass over grouped frames --
    # k_sweep_cand -- has no reduce kernel behind it)
    return red[hint["first"]:hint["first"] + hint["count"]] if hint["kernel"] == "k_pass_pre" else []


def counters_of(path, rows_sel):
    ids = {r["Dispatch_Id"] for r in rows_sel}
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Dispatch_Id"] in ids:
            per.setdefault(r["Dispatch_Id"], {})
            per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    vals = list(per.values())
    return {k: sum(x[k] for x in vals) / len(vals) for k in vals[0]} if vals else {}


def counters(path, hint):
    per, order = {}, []
    for r in csv.DictReader(open(path)):
        if not is_dominant(r["Kernel_Name"], hint["kernel"]):
            continue
        d = r["Dispatch_Id"]
        if d not in per:
            per[d] = {}
            order.append(d)
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    sel = [per[d] for d in order][hint["first"]:hint["first"] + hint["count"]]
    return {k: sum(x[k] for x in sel) / len(sel) for k in sel[0]}, sel


def main():
    os.makedirs(DST, exist_ok=True)
    global M
    b = bench_line("kt")
    hint = b["roofline"]["trace_dispatches"]
    M = b["config"].get("codebook_size", 1024)
    global T
    T = b["config"].get("frames_per_gpu", T)
    L = b["config"]["passes_per_level"]
    if hint["kernel"] == "k_pass_pre" and hint["count"] != b["steps"]:
        L = max(1, hint["count"] * L // b["steps"])  # (some passes of the level ran the plain sweep: M = 256)
    kt = newest(f"{SRC}/{TAG}_kt/*/*_kernel_trace.csv")
    rows = timed_rows(kt, hint)
    d = [ms(r) for r in rows]
    by_pos = [[d[i] for i in range(len(d)) if i % L == k] for k in range(L)]
    out = {
        "kernel": rows[0]["Kernel_Name"].split("(")[0][:80],
        "dispatches": len(d), "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
        "avg_ms_by_pass_of_level": [sum(x) / len(x) for x in by_pos if x],
        "note": f"timed region = {len(d) // L} repetitions of the real M={M} level ({L} passes: the first seeded with the parents' sums, "
                "the others incremental); bench.py's HIP-event average over the same launches is bench_kernel_ms",
        "bench_kernel_ms": b["roofline"]["kernel_ms"], "bench_ms_per_step": b["ms_per_step"], "bench_value": b["value"],
        # (no register counts here: rocprofv3's VGPR_Count column is not the wave's allocation on gfx950 -- it printed 128 for a
        # kernel whose descriptor says 256; tests/test_isa_guards.py reads the counts from the compiled ISA and pins them)
        "frames_per_launch": T,
        "lds_bytes": rows[0].get("LDS_Block_Size"), "grid": rows[0].get("Grid_Size"), "workgroup": rows[0].get("Workgroup_Size"),
        "kernel_sources_sha16": kernel_sources_sha16(),
        "source": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --steps 21 "
                  f"--warmup 3; timed dispatches = {hint['kernel']} launches [{hint['first']}, {hint['first'] + hint['count']})",
    }
    red = reduce_rows(kt, hint)
    if red:
        rd = [ms(r) for r in red]
        out["accumulate_kernel"] = {
            "kernel": "k_reduce_records<37>: the launch behind each of the timed sweeps (records sorted by cell in LDS, rows summed "
                      "in registers)",
            "dispatches": len(rd), "avg_ms": sum(rd) / len(rd),
            "avg_ms_by_pass_of_level": [sum(rd[i] for i in range(len(rd)) if i % L == k) / max(1, len([i for i in range(len(rd)) if i % L == k])) for k in range(L)],
            "bench_accumulate_kernel_ms": b["roofline"].get("accumulate_kernel_ms"),
            "vgpr": red[0].get("VGPR_Count"), "lds_bytes": red[0].get("LDS_Block_Size"), "grid": red[0].get("Grid_Size"),
            "workgroup": red[0].get("Workgroup_Size"),
        }
    sfx = "" if M == 1024 else f"_M{M}"
    json.dump(out, open(f"{DST}/{TAG}_pass_kernel{sfx}.json", "w"), indent=1)
    shutil.copy(newest(f"{SRC}/{TAG}_kt/*/*_kernel_stats.csv"), f"{DST}/{TAG}_kernel_stats{sfx}.csv")
    json.dump(b, open(f"{DST}/{TAG}_bench_under_trace{sfx}.json", "w"), indent=1)
    print(json.dumps(out, indent=1))

    # ---- PMC traffic ------------------------------------------------------------------------------------------
    try:
        hf, hw = bench_line("fetch")["roofline"]["trace_dispatches"], bench_line("write")["roofline"]["trace_dispatches"]
        f, fsel = counters(newest(f"{SRC}/{TAG}_fetch/*/*_counter_collection.csv"), hf)
        w, wsel = counters(newest(f"{SRC}/{TAG}_write/*/*_counter_collection.csv"), hw)
        prefiltered = hint["kernel"] in ("k_pass_pre", "k_sweep_cand")
        tj = {
            "kernel": out["kernel"] + f" at M={M}, {T} frames per launch (bench.py --steps 6: two repetitions of the {L}-pass level)",
            "frames_per_launch": T,
            "FETCH_SIZE_KB_raw": f["FETCH_SIZE"], "WRITE_SIZE_KB_raw": w["WRITE_SIZE"],
            "FETCH_SIZE_KB_per_step": [x["FETCH_SIZE"] for x in fsel], "WRITE_SIZE_KB_per_step": [x["WRITE_SIZE"] for x in wsel],
            "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled "
                          "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for atomics / 16-B stores",
            "fetch_bytes": f["FETCH_SIZE"] * 2048, "write_bytes": w["WRITE_SIZE"] * 1024,
            "hbm_bytes_per_launch": f["FETCH_SIZE"] * 2048 + w["WRITE_SIZE"] * 1024,
            "algorithmic_bytes_per_launch": 306 * T,
            "kernel_sources_sha16": kernel_sources_sha16(),
            "note": "average over the timed launches (full-accumulate and incremental passes of the level mixed as in the bench)",
        }
        try:  # the accumulate kernel's own traffic (row re-reads, records, flush atomics)
            ff = newest(f"{SRC}/{TAG}_fetch/*/*_counter_collection.csv")
            fw = newest(f"{SRC}/{TAG}_write/*/*_counter_collection.csv")
            rf = counters_of(ff, reduce_rows(newest(f"{SRC}/{TAG}_fetch/*/*_kernel_trace.csv"), hf))
            rw = counters_of(fw, reduce_rows(newest(f"{SRC}/{TAG}_write/*/*_kernel_trace.csv"), hw))
            if rf and rw:
                tj["accumulate_kernel"] = {"kernel": "k_reduce_records", "fetch_bytes": rf["FETCH_SIZE"] * 2048,
                                           "write_bytes": rw["WRITE_SIZE"] * 1024,
                                           "note": "per launch, averaged over the timed launches; same corrections"}
        except (IndexError, FileNotFoundError, KeyError) as ex:
            print("no traffic of the accumulate kernel:", ex)
        json.dump(tj, open(f"{DST}/{TAG}_traffic{sfx}{'' if prefiltered else '_noprefilter'}.json", "w"), indent=1)
        print("fetch raw KB", f["FETCH_SIZE"], "write raw KB per step", [round(x["WRITE_SIZE"]) for x in wsel])
    except (IndexError, FileNotFoundError, KeyError) as ex:
        print("no PMC traffic passes:", ex)

    # ---- SQ counters --------------------------------------------------------------------------------------------
    try:
        h1, h2 = bench_line("sq1")["roofline"]["trace_dispatches"], bench_line("sq2")["roofline"]["trace_dispatches"]
        c1, _ = counters(newest(f"{SRC}/{TAG}_sq1/*/*_counter_collection.csv"), h1)
        c2, _ = counters(newest(f"{SRC}/{TAG}_sq2/*/*_counter_collection.csv"), h2)
        d1 = [ms(r) for r in timed_rows(newest(f"{SRC}/{TAG}_sq1/*/*_kernel_trace.csv"), h1)]
        d2 = [ms(r) for r in timed_rows(newest(f"{SRC}/{TAG}_sq2/*/*_kernel_trace.csv"), h2)]
        c = {**c1, **c2}
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        sq = {"kernel": out["kernel"], "counters": c, "kernel_ms_under_pmc": [sum(d1) / len(d1), sum(d2) / len(d2)],
              "kernel_cycles_under_pmc": cyc, "clock_GHz_under_pmc": cyc / (sum(d2) / len(d2) * 1e-3) / 1e9,
              "mfma_pipe_busy_fraction": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
              "other_valu_instructions_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"],
              "kernel_sources_sha16": kernel_sources_sha16(),
              "note": "busy cycles summed over the 1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; averages over the "
                      "timed launches of two PMC-only runs"}
        if hint["kernel"] == "k_pass_pre":
            f16 = (T // 32) * (M // 32) * 15
            sq["expected_f16_mfma_instructions"] = f16
            sq["fp64_mfma_instructions"] = c["SQ_INSTS_MFMA"] - f16
        if hint["kernel"] == "k_sweep_cand":
            ks = b["roofline"].get("ksteps_per_pair", 15.0)
            sq["expected_f16_mfma_instructions"] = (T // 32) * (M // 32) * ks
            sq["ksteps_per_pair_from_the_bench_line"] = ks
            sq["note_mfma"] = "SQ_INSTS_MFMA against (frames / 32) x (codewords / 32) x k-steps per pair: the flagged share varies a little from pass to pass"
        json.dump(sq, open(f"{DST}/{TAG}_sq_counters{sfx}.json", "w"), indent=1)
        print({k: sq[k] for k in ("kernel_ms_under_pmc", "clock_GHz_under_pmc", "mfma_pipe_busy_fraction",
                                  "other_valu_instructions_per_mfma")})
    except (IndexError, FileNotFoundError, KeyError) as ex:
        print("no SQ passes:", ex)


if __name__ == "__main__":
    main()
