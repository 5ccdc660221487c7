"""Top rows of a rocprofv3 kernel_stats.csv with total / average times. usage: top_kernels.py stats.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e9:.3f} s over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:58]
    print(f"{name:58s} calls={r['Calls']:>7s} total_ms={float(r['TotalDurationNs']) / 1e6:9.1f} avg_us={float(r['AverageNs']) / 1e3:9.1f}")
