#!/bin/bash
# usage (on the GPU box): bash tools/lut_trace.sh "m n bits M" ...   -- rocprofv3 kernel durations per configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/lt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o lt -- python3 $R/tools/bench_lut_trace.py $cfg > /tmp/lt.log 2>&1 || { tail -5 /tmp/lt.log; exit 1; }
  echo "== $cfg"
  python3 $R/tools/print_kernel_stats.py "$(find /tmp/lt -name 'lt_kernel_stats.csv' | head -1)" 3
done
