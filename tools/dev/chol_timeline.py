#!/usr/bin/env python3
"""developer: kernel timeline of the LAST ganq_cholesky of a rocprofv3 --kernel-trace csv (start offsets, durations, stream gaps)
usage: chol_timeline.py kernel_trace.csv [max_rows]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "chol_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("ganq::", "") for r in rows]
ends = [i for i, nm in enumerate(names) if nm.startswith("chol_zero_upper")]
a = ends[-2] + 1 if len(ends) > 1 else 0
b = ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev_diag = None
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    extra = ""
    if names[i].startswith("chol_diag"):
        if prev_diag is not None: extra = f"  (step {(s - prev_diag) / 1e3:.1f} us)"
        prev_diag = s
    if i - a < mx: print(f"{names[i]:24s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  grid {rows[i].get('Grid_Size', '?'):>8s}{extra}")
print(f"span {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
