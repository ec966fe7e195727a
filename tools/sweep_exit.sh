#!/bin/bash
# exit rule of the traversal phase: PT_EXIT_LANES x PT_EXIT_MUL on C5 (8 spp) and on book-1 (knob_sweep, 128 spp)
cd "$(dirname "$0")/.."
for ex in 48 64; do for mul in 4 6 8 12 16 32; do
echo "C5 exit=$ex mul=$mul"
PT_EXIT_LANES=$ex PT_EXIT_MUL=$mul timeout -k 10 100 python tools/c5_spp.py 8 || exit 1
done; done
timeout -k 10 300 python tools/knob_sweep.py PT_EXIT_LANES=8,16,32,64 PT_EXIT_MUL=4,8,16
