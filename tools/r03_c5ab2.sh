#!/bin/bash
# C5 after -fno-slp-vectorize: waves per SIMD and stack levels in LDS once more
set -u
mkdir -p gpurun_out/r03i
L=gpurun_out/r03i/c5ab2.log
run() { echo "== $*" | tee -a $L; env "$@" 2>&1 | grep -E "Msamples" | sed 's/"checksum".*//' | tee -a $L; }
run timeout -k 10 200 python tools/ab_c5.py --spp 16 libpt_hip.so
run timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_hbm7.so
run PT_WIDE_STACK=8 timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_hbm7.so
run timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_hbm5.so
run PT_WIDE_STACK=8 timeout -k 10 200 python tools/ab_c5.py --spp 16 libpt_hip.so
run timeout -k 10 200 python tools/ab_c5.py --spp 16 libpt_hip.so
