#!/bin/bash
# usage (on the GPU box): bash tools/collect_round.sh <tag>   e.g. r03_v2
# One pass over everything the round's docs quote: bench line, rocprofv3 kernel stats of the same command, PMC traffic of
# the loop's kernels, model-level runs, per-shape quantize() times, LUT GEMM table.  Writes gpurun_out/<tag>_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-rXX}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "[collect] bench"; python3 bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err || tail -3 $O/${T}_bench.err
echo "[collect] kernel stats"; bash tools/bench_stats.sh 30 > $O/${T}_bench_stats.txt 2>&1; cp $O/bench_kernel_stats.csv $O/${T}_bench_kernel_stats.csv
echo "[collect] pmc traffic"; bash tools/pmc_collect.sh $O/${T}_pmc_traffic.json > $O/${T}_pmc.txt 2>&1
echo "[collect] models"; python3 tools/quantize_model_bench.py --arch llama-3.2-1b > $O/${T}_model_llama32_1b.json 2> $O/${T}_model.err
python3 tools/quantize_model_bench.py --arch opt-125m > $O/${T}_model_opt125m.json 2>> $O/${T}_model.err
echo "[collect] shapes"; python3 tools/time_quantize_shapes.py > $O/${T}_quantize_shapes.txt 2>&1
echo "[collect] lut gemm"; python3 tools/bench_lut_gemm.py --shapes 4096x4096,14336x4096,4096x14336,2048x8192 --M 128,512,1024,2048,4096 > $O/${T}_lut_gemm.jsonl 2> $O/${T}_lut_gemm.err
echo "[collect] hessian / kmeans"; python3 tools/time_hessian.py > $O/${T}_hessian.txt 2>&1; python3 tools/time_kmeans.py 4096x4096x16 4096x4096x8 4096x11008x16 1024x2048x16 8192x2048x16 768x3072x16 > $O/${T}_kmeans.txt 2>&1
echo "[collect] dense gemm"; (cd tools && python3 bench_gemm_h16.py --M 1024,2048,4096 > $O/${T}_gemm_h16.jsonl 2>&1)
echo "[collect] cholesky / hessian paths"; python3 tools/time_cholesky.py > $O/${T}_cholesky.txt 2>&1; python3 tools/dev/hess_w4_ab.py > $O/${T}_hessian_paths.txt 2>&1
echo "[collect] done"
