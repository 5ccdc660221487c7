"""Trim a rocprofv3 *_kernel_stats.csv to something readable: kernel names cut at the first '(' / 90 chars."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.reader(open(src)))
out = [rows[0]]
for r in rows[1:]:
    name = r[0].replace("(anonymous namespace)::", "").split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    r[0] = name[:90]
    out.append(r)
csv.writer(open(dst, "w")).writerows(out[:25])
