#!/bin/bash
# usage (on the GPU box): bash tools/loop_trace.sh [rows]  -- kernel timeline of one ganq_run_layer of bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o lt -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-lut --no-opt125m --no-llama --no-stress --no-tiny > /tmp/lt.log 2>&1 || { tail -5 /tmp/lt.log; exit 1; }
python3 $R/tools/loop_timeline.py "$(find /tmp/lt -name 'lt_kernel_trace.csv' | head -1)" ${1:-60}
