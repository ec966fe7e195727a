#!/bin/bash
set -u
mkdir -p gpurun_out/r03b
run() { echo "== $*" | tee -a gpurun_out/r03b/sah.log; env "$@" timeout -k 10 120 python tools/ab.py --spp ${SPP:-256} --reps 3 libpt_hip.so 2>/dev/null | tee -a gpurun_out/r03b/sah.log || exit 1; }
run PT_SAH_SWEEP=512
run PT_SAH_SWEEP=512 PT_LEAF_MAX=2
run PT_SAH_SWEEP=512 PT_LEAF_MAX=3
run PT_LEAF_MAX=3
run PT_LEAF_MAX=2
run PT_SAH_SWEEP=512 PT_LEAF_MAX=1
run PT_NONE=1
