"""Timeline of a rocprofv3 --kernel-trace CSV: per dispatch its duration and the idle gap in front of it.
usage: python3 tools/trace_gaps.py <kernel_trace.csv> [first_dispatch [count]]   (short kernel names)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
def short(n):
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_[A-Za-z]+)", n)
    return m.group(1) if m else n[:40]
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if first <= i < first + count:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{i:5d} {short(r['Kernel_Name']):28s} dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us  grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']}")
    prev_end = e if prev_end is None else max(prev_end, e)
