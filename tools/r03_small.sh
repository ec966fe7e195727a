#!/bin/bash
set -u
mkdir -p gpurun_out/r03b
for sc in 3 4; do 
echo "== scene $sc $*" | tee -a gpurun_out/r03b/small.log
timeout -k 10 120 python tools/ab.py --scene $sc --flags 0 --spp 256 --reps 3 "$@" 2>/dev/null | tee -a gpurun_out/r03b/small.log || exit 1
done
