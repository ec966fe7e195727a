#!/bin/bash
# knob sweep of the tree-in-LDS kernel after the leaf blocks: one knob at a time around the defaults (book-1 1080p, 256 spp)
set -u
mkdir -p gpurun_out/r03b
run() { echo "== $*" | tee -a gpurun_out/r03b/sweep.log; env "$@" timeout -k 10 120 python tools/ab.py --spp ${SPP:-256} --reps 3 libpt_hip.so 2>/dev/null | tee -a gpurun_out/r03b/sweep.log || exit 1; }
run PT_NONE=1
for v in 8 12 16 24 28 32; do run PT_LEAF_BATCH=$v; done
for v in 6 8 12 14; do run PT_EXIT_LANES=$v; done
for v in 20 24 32 36; do run PT_REFILL_MIN=$v; done
run PT_NONE=1
