#!/bin/bash
# usage (on the GPU box): bash tools/bench_stats.sh [rows] -- bench.py JSON line + rocprofv3 kernel stats of the same command
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bs
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs -o bs -- python3 $R/bench.py --no-cpu-baseline --no-lut --no-opt125m --no-llama --no-stress --no-tiny > $R/gpurun_out/bench_stats.json 2>/tmp/bs.err || { tail -5 /tmp/bs.err; exit 1; }
cp "$(find /tmp/bs -name 'bs_kernel_stats.csv' | head -1)" $R/gpurun_out/bench_kernel_stats.csv
python3 $R/tools/top_kernels.py $R/gpurun_out/bench_kernel_stats.csv ${1:-24}
tail -1 $R/gpurun_out/bench_stats.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels'].get('t_prepare_kernels'))"
