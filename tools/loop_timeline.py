"""Timeline of the last ganq_run_layer of a rocprofv3 --kernel-trace: kernels in order with start offsets, durations and
the idle gap before each.  usage: loop_timeline.py kernel_trace.csv [max_rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ganq::", "") for r in rows]
# last launch group that starts with l_pack_kernel (one per run_layer)
starts = [i for i, nm in enumerate(names) if nm.startswith("l_pack_kernel")]
a = starts[-1]
b = len(rows)
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = gap = 0
out = []
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    if s - prev_end > 2_000_000:  # left the layer (next phase of the script)
        break
    g = max(0, s - prev_end)
    busy += e - s
    gap += g
    out.append((names[i][:44], (s - t0) / 1e3, (e - s) / 1e3, g / 1e3))
    prev_end = max(prev_end, e)
mx = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for nm, st, du, g in out[:mx]:
    print(f"{nm:44s} start {st:9.1f} us  dur {du:8.1f} us  gap {g:6.1f} us")
print(f"launches {len(out)}  span {(prev_end - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {gap / 1e3:.1f} us")
