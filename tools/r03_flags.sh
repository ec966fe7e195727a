#!/bin/bash
# A/B of compiler-flag variants (make variant NAME=.. DEFS=..) on book-1 512 spp and C5-like scene 2
set -u
mkdir -p gpurun_out/r03h
LIBS="libpt_hip.so $(cd pathtracer_amd/lib && ls ab/libpt_hip_v_*.so | tr '\n' ' ') libpt_hip.so"
timeout -k 10 600 python tools/ab.py --spp 512 --reps 3 $LIBS 2>/dev/null | tee gpurun_out/r03h/flags.log || exit 1
