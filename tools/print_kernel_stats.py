"""Print name / calls / average ns of the top rows of a rocprofv3 kernel_stats.csv (names cut short)."""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
h = rows[0]
ni, ci, ai, ti = h.index("Name"), h.index("Calls"), h.index("AverageNs"), h.index("TotalDurationNs")
print(f"  all kernels: {sum(float(r[ti]) for r in rows[1:]) / 1e6:.1f} ms")
for r in rows[1:1 + top]:
    print(f"  {r[ni].replace('(anonymous namespace)::', '').split('(')[0][:70]:70s} calls={r[ci]:>6s} avg_us={float(r[ai]) / 1e3:9.2f} total_ms={float(r[ti]) / 1e6:9.2f}")
