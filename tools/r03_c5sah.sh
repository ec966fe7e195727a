#!/bin/bash
# C5: the exact SAH sweep at the bottom of the big tree (ranges of up to N spheres), visits per segment / time / upload
set -u
mkdir -p gpurun_out/r03t
L=gpurun_out/r03t/c5sah.log
run() { echo "== $*" | tee -a $L; env "$@" timeout -k 10 300 python tools/ab_c5.py --spp 16 libpt_hip.so 2>/dev/null | sed 's/"checksum".*//' | tee -a $L; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run PT_NONE=1
run PT_SAH_SWEEP=32
run PT_SAH_SWEEP=128
run PT_SAH_SWEEP=512
