#!/bin/bash
# after -fno-slp-vectorize: the few-sphere loop's filter plain against packed, and the tree kernel's knobs once more
set -u
mkdir -p gpurun_out/r03h
L=gpurun_out/r03h/sweep3.log
for sc in 3 4; do timeout -k 10 200 python tools/ab.py --spp 256 --scene $sc libpt_hip.so ab/libpt_hip_v_fewplain.so libpt_hip.so ab/libpt_hip_v_fewplain.so 2>/dev/null | tee -a $L || exit 1; done
run() { echo "== $*" | tee -a $L; env "$@" timeout -k 10 120 python tools/ab.py --spp 256 --reps 3 libpt_hip.so 2>/dev/null | tee -a $L || exit 1; }
run PT_NONE=1
for v in 8 12 24 32; do run PT_LEAF_BATCH=$v; done
for v in 6 8 12 14; do run PT_EXIT_LANES=$v; done
for v in 20 24 32 36; do run PT_REFILL_MIN=$v; done
for c in 16 32 48; do run PT_CARRY_MAX=$c; done
run PT_NONE=1
