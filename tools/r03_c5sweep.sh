#!/bin/bash
# C5 knobs once more on the end-of-round build (7 waves per SIMD, 8 stack levels in LDS), one at a time around the defaults, 16 spp
set -u
mkdir -p gpurun_out/r03l
L=gpurun_out/r03l/c5sweep.log
run() { echo "== $*" | tee -a $L; env "$@" timeout -k 10 200 python tools/ab_c5.py --spp 16 libpt_hip.so 2>/dev/null | sed 's/"checksum".*//' | tee -a $L; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run PT_NONE=1
for v in 4 12 16; do run PT_LEAF_BATCH=$v; done
for v in 32 40 56 64; do run PT_EXIT_LANES=$v; done
for v in 4 5 8; do run PT_EXIT_MUL=$v; done
for v in 4 12 16; do run PT_REFILL_MIN=$v; done
for v in 32 64; do run PT_CARRY_MAX=$v; done
for v in 6 10 12; do run PT_WIDE_STACK=$v; done
run PT_NONE=1
