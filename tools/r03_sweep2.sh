#!/bin/bash
# chunk schedule re-swept on the round-3 kernel (book-1 1080p, 512 spp): PT_GUIDE x PT_CHUNK_MIN, then carry / refill
set -u
mkdir -p gpurun_out/r03b
run() { echo "== $*" | tee -a gpurun_out/r03b/sweep2.log; env "$@" timeout -k 10 120 python tools/ab.py --spp ${SPP:-512} --reps 3 libpt_hip.so 2>/dev/null | tee -a gpurun_out/r03b/sweep2.log || exit 1; }
run PT_NONE=1
for g in 12 16 32 48; do run PT_GUIDE=$g; done
for c in 2 8 16; do run PT_CHUNK_MIN=$c; done
for c in 8 16 32 48; do run PT_CARRY_MAX=$c; done
run PT_NONE=1
