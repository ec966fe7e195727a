#!/bin/bash
# End-of-round measurement set (one gpurun call): tools/final_measure.sh <tag>; everything lands under gpurun_out/final_<tag>/
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/gpu_tests.log 2>&1 || { tail -5 $OUT/gpu_tests.log; exit 1; }
tail -1 $OUT/gpu_tests.log
# BASELINE.json's C3 / C4 / C5 at their full sizes (the test prints one line per configuration)
timeout -k 10 300 python -u -m pytest tests/test_progressive.py -m gpu -x -q -s -k full_size --timeout 200 --timeout-method thread 2>&1 | grep -oE "C[0-9]_[a-z0-9_]+: .*" > $OUT/full_size.txt || { echo "full-size test failed"; exit 1; }
# the bench lines FIRST, on a box that has not yet run forty profiler passes (the same build measured 3 % slower behind them)
timeout -k 10 300 python bench.py --scene 4 --spp-per-gpu 256 --steps 20 --warmup 5 > $OUT/c2_bench.json 2> $OUT/c2_bench.err || { tail -3 $OUT/c2_bench.err; exit 1; }
timeout -k 10 300 python bench.py --scene 3 --width 3840 --height 2160 --spp-per-gpu 128 --steps 20 --warmup 5 > $OUT/c4_bench.json 2> $OUT/c4_bench.err || { tail -3 $OUT/c4_bench.err; exit 1; }
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
timeout -k 10 300 python bench.py --scene 5 --spheres 1000000 --width 3840 --height 2160 --spp-per-gpu 16 --cpu-spp 0 --steps 20 --warmup 5 > $OUT/c5_bench.json 2> $OUT/c5_bench.err || { tail -3 $OUT/c5_bench.err; exit 1; }
bash tools/profile.sh $TAG --steps 3 --warmup 1 > $OUT/profile.log 2>&1 || { tail -3 $OUT/profile.log; exit 1; }
bash tools/profile_c5.sh ${TAG}_c5 > $OUT/profile_c5.log 2>&1 || { tail -3 $OUT/profile_c5.log; exit 1; }
# C2 (materials + DOF, 1080p, 256 spp) and C4 (motion blur, 4K, 128 spp): the same passes, each under its own traffic key
PT_TRAFFIC_KEY=scene4_1920x1080_spp256_d50 bash tools/profile.sh ${TAG}_c2 --scene 4 --spp-per-gpu 256 --steps 3 --warmup 1 > $OUT/profile_c2.log 2>&1 || { tail -3 $OUT/profile_c2.log; exit 1; }
PT_TRAFFIC_KEY=scene3_3840x2160_spp128_d50 bash tools/profile.sh ${TAG}_c4 --scene 3 --width 3840 --height 2160 --spp-per-gpu 128 --steps 3 --warmup 1 > $OUT/profile_c4.log 2>&1 || { tail -3 $OUT/profile_c4.log; exit 1; }
timeout -k 10 300 python tools/bench_configs.py > $OUT/configs.jsonl 2>/dev/null || exit 1
timeout -k 10 100 python tools/c5_spp.py 4 16 64 >> $OUT/configs.jsonl 2>/dev/null || exit 1
timeout -k 10 600 python tools/report.py > $OUT/report.json 2> $OUT/report.err || { tail -3 $OUT/report.err; exit 1; }
timeout -k 10 100 python tools/phases.py 1 128 > $OUT/phases_book1.txt 2>&1 || exit 1
PH_W=3840 PH_H=2160 PH_SPHERES=1000000 timeout -k 10 200 python tools/phases.py 5 8 > $OUT/phases_c5.txt 2>&1 || exit 1
PT_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --cpu-spp 0 --no-brute --steps 20 --warmup 5 > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err || { tail -3 $OUT/bench_forcedist.err; exit 1; }
echo all done
