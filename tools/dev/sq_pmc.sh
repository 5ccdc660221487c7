#!/bin/bash
# developer (GPU box): SQ counters of the kernels whose name contains <filter>, averaged per launch
# usage: bash tools/dev/sq_pmc.sh <filter> python3 script.py args...
R=${GRAFT_REPO_ROOT:-$(pwd)}
FILT=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  rm -rf /tmp/sqp
  ( cd $R && rocprofv3 --pmc $set --output-format csv -d /tmp/sqp -o p -- "$@" > /tmp/sqp.log 2>&1 ) || { tail -3 /tmp/sqp.log; exit 1; }
  FILT=$FILT python3 - <<'PY'
import csv, glob, collections, os
filt = os.environ["FILT"]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/sqp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (s, c) in sorted(acc.items()):
    print(f"{kn:40s} {cn:28s} {s / c:16.0f}  ({c} launches)")
PY
done
