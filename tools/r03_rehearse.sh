#!/bin/bash
# 2-rank rehearsals of bench.py's own launcher on a one-GPU box (both ranks share the GPU; gloo through host memory)
set -u
mkdir -p gpurun_out/r03a
export PT_BENCH_BACKEND=gloo PT_BENCH_WATCHDOG=100
timeout -k 10 200 python bench.py --gpus 2 --steps 3 --warmup 1 --spp-per-gpu 64 > gpurun_out/r03a/bench_2rank_gloo.json 2> gpurun_out/r03a/bench_2rank_gloo.err || { tail -40 gpurun_out/r03a/bench_2rank_gloo.err; exit 1; }
tail -1 gpurun_out/r03a/bench_2rank_gloo.json
timeout -k 10 200 python bench.py --gpus 2 --steps 3 --warmup 1 --spp-total 128 > gpurun_out/r03a/bench_2rank_gloo_strong.json 2> gpurun_out/r03a/bench_2rank_gloo_strong.err || { tail -40 gpurun_out/r03a/bench_2rank_gloo_strong.err; exit 1; }
tail -1 gpurun_out/r03a/bench_2rank_gloo_strong.json
