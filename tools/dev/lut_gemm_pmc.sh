#!/bin/bash
# developer (GPU box): SQ counters of the LUT GEMM kernels at one shape; usage: bash tools/dev/lut_gemm_pmc.sh m n M
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pipe in 0 1; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"; do
    rm -rf /tmp/lgp
    rocprofv3 --pmc $set --output-format csv -d /tmp/lgp -o p -- python3 $R/tools/dev/lut_gemm_once.py $1 $2 $3 $pipe > /tmp/lgp.log 2>&1 || { tail -3 /tmp/lgp.log; exit 1; }
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/lgp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lut_gemm" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (s, c) in sorted(acc.items()):
    print(f"pipe=$pipe {kn:24s} {cn:28s} {s / c:16.0f}")
PY
  done
done
