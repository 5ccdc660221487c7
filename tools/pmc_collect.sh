#!/bin/bash
# usage (on the GPU box): bash tools/pmc_collect.sh <out.json> -- HBM-side bytes per kernel of one bench step.
# Two passes (FETCH_SIZE and WRITE_SIZE do not fit the TCC counter slots together); counters only, no tracing.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc_traffic.json}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-lut --no-opt125m --no-llama --no-stress --no-tiny > /tmp/pmc_f.log 2>&1 || { tail -5 /tmp/pmc_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-lut --no-opt125m --no-llama --no-stress --no-tiny > /tmp/pmc_w.log 2>&1 || { tail -5 /tmp/pmc_w.log; exit 1; }
python3 $R/tools/pmc_parse.py $OUT /tmp/pmc_f /tmp/pmc_w
