#!/bin/bash
# C5 threshold sweep (leaf batch x exit lanes, refill threshold) at 8 spp: tools/sweep_c5.sh > gpurun_out/sweep_c5.txt
set -e
cd "$(dirname "$0")/.."
for lb in 4 8 12; do for ex in 40 48 56; do
echo "leaf_batch=$lb exit=$ex"
PT_LEAF_BATCH=$lb PT_EXIT_LANES=$ex timeout -k 10 100 python tools/c5_spp.py 8
done; done
for rf in 16 20 28 36; do echo "refill=$rf"; PT_REFILL_MIN=$rf timeout -k 10 100 python tools/c5_spp.py 8; done
