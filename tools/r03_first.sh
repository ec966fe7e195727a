#!/bin/bash
# round 3, first GPU call: the whole GPU suite, the bench line, the two forced-collective rehearsals, a 2-rank launcher rehearsal
set -u
mkdir -p gpurun_out/r03a
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03a/pytest.log 2>&1 || { tail -30 gpurun_out/r03a/pytest.log; exit 1; }
tail -3 gpurun_out/r03a/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err || { tail -20 gpurun_out/r03a/bench.err; exit 1; }
cat gpurun_out/r03a/bench.json
PT_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-spp 0 --no-brute > gpurun_out/r03a/bench_forcedist.json 2> gpurun_out/r03a/bench_forcedist.err || { tail -20 gpurun_out/r03a/bench_forcedist.err; exit 1; }
cat gpurun_out/r03a/bench_forcedist.json
PT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --spp-per-gpu 64 > gpurun_out/r03a/bench_2rank_gloo.json 2> gpurun_out/r03a/bench_2rank_gloo.err || { tail -20 gpurun_out/r03a/bench_2rank_gloo.err; exit 1; }
cat gpurun_out/r03a/bench_2rank_gloo.json
PT_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --spp-total 128 > gpurun_out/r03a/bench_2rank_gloo_strong.json 2> gpurun_out/r03a/bench_2rank_gloo_strong.err || { tail -20 gpurun_out/r03a/bench_2rank_gloo_strong.err; exit 1; }
cat gpurun_out/r03a/bench_2rank_gloo_strong.json
