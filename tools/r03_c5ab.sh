#!/bin/bash
set -u
mkdir -p gpurun_out/r03b
run() { echo "== $*" | tee -a gpurun_out/r03b/c5ab.log; env "$@" 2>&1 | grep -E "Msamples|render mode" | sort -u | sed 's/"checksum".*//' | tee -a gpurun_out/r03b/c5ab.log; }
run PT_VERBOSE=1 timeout -k 10 200 python tools/ab_c5.py --spp 16 libpt_hip.so
run PT_VERBOSE=1 timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_park6.so
run PT_VERBOSE=1 PT_WIDE_STACK=8 timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_park6.so
run PT_VERBOSE=1 PT_WIDE_STACK=8 timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_park7.so
run PT_VERBOSE=1 PT_WIDE_STACK=8 timeout -k 10 200 python tools/ab_c5.py --spp 16 ab/libpt_hip_hbm7.so
run PT_VERBOSE=1 PT_WIDE_STACK=8 timeout -k 10 200 python tools/ab_c5.py --spp 16 libpt_hip.so
