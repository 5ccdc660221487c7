#!/bin/bash
# usage (on the GPU box): bash tools/model_stats.sh [arch] -- whole-model quantization (tools/quantize_model_bench.py) under
# rocprofv3 --kernel-trace --stats: where the GPU time of a model-level run goes, kernel by kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=${1:-llama-3.2-1b}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ms
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ms -o ms -- python3 $R/tools/quantize_model_bench.py --arch $A > $R/gpurun_out/model_stats_$A.json 2>/tmp/ms.err || { tail -5 /tmp/ms.err; exit 1; }
cp "$(find /tmp/ms -name 'ms_kernel_stats.csv' | head -1)" $R/gpurun_out/model_kernel_stats_$A.csv
python3 $R/tools/print_kernel_stats.py $R/gpurun_out/model_kernel_stats_$A.csv 30
tail -1 $R/gpurun_out/model_stats_$A.json
