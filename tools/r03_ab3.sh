#!/bin/bash
# A/B over several scenes: usage tools/r03_ab3.sh "scene list" lib...
set -u
mkdir -p gpurun_out/r03b
SC="$1"; shift
for sc in $SC; do
echo "== scene $sc spp ${SPP:-256}" | tee -a gpurun_out/r03b/ab3.log
timeout -k 10 200 python tools/ab.py --scene $sc --spp ${SPP:-256} --reps 3 "$@" 2>/dev/null | tee -a gpurun_out/r03b/ab3.log || exit 1
done
