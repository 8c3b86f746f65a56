#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py`: the timed steps are the LAST `steps` dispatches of the
sweep kernel (the earlier ones belong to the untimed ladder 2..512 and the warm-up).
usage: tools/summarize_trace.py <kernel_trace.csv> <steps> [out.json]"""
import csv
import json
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = [r for r in csv.DictReader(open(path)) if "k_pass_mfma<37, 2" in r["Kernel_Name"]]
    last = rows[-steps:]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in last]
    out = {
        "kernel": last[0]["Kernel_Name"].split("(")[0],
        "dispatches": len(d),
        "avg_ms": sum(d) / len(d),
        "min_ms": min(d),
        "max_ms": max(d),
        "grid": last[0].get("Grid_Size"),
        "workgroup": last[0].get("Workgroup_Size"),
        "vgpr": last[0].get("VGPR_Count"),
        "accum_vgpr": last[0].get("Accum_VGPR_Count"),
        "lds_bytes": last[0].get("LDS_Block_Size"),
        "source": path,
    }
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(s + "\n")


if __name__ == "__main__":
    main()
