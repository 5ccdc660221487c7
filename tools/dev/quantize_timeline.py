#!/usr/bin/env python3
"""developer: kernel timeline of the LAST quantize() of a rocprofv3 --kernel-trace csv of tools/time_quantize.py: everything from the last
prologue_rowstats_kernel on -- start offsets, durations, idle gaps (all streams merged), and the sums per phase
usage: quantize_timeline.py kernel_trace.csv [min_us_to_print]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ganq::", "")[:46]
starts = [i for i, r in enumerate(rows) if nm(r).startswith("prologue_rowstats_kernel")]
a = starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
busy_end = t0
gap_total = 0.0
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(0, s - busy_end) / 1e3
    gap_total += g
    d = (e - s) / 1e3
    if d >= thr or g >= 5.0:
        print(f"{nm(r):46s} start {(s - t0) / 1e3:9.1f} us  dur {d:8.1f} us  idle before {g:6.1f} us")
    busy_end = max(busy_end, e)
print(f"span {(busy_end - t0) / 1e3:.1f} us, idle (no kernel on any stream) {gap_total:.1f} us")
