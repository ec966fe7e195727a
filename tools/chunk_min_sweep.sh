#!/bin/bash
# smallest sample chunk (PT_CHUNK_MIN) against frame size and sample count
cd "$(dirname "$0")/.."
for cfg in "800 450 5 8" "800 450 5 32" "800 450 5 128" "800 450 50 32" "1920 1080 50 8" "1920 1080 50 16" "1920 1080 50 64" "3840 2160 50 8"; do
set -- $cfg
echo "== $1x$2 depth $3 spp $4"
C1_W=$1 C1_H=$2 C1_DEPTH=$3 C1_SPP=$4 timeout -k 10 120 python tools/c1_sweep.py PT_CHUNK_MIN=2,4,8 2>/dev/null | grep '^{' || exit 1
done
