#!/bin/bash
# A/B on the GPU box: GPU suite on the working tree's library first (parity before speed), then tools/ab.py over the libraries named
set -u
mkdir -p gpurun_out/r03b
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r03b/pytest.log 2>&1 || { tail -40 gpurun_out/r03b/pytest.log; exit 1; }
tail -2 gpurun_out/r03b/pytest.log
fi
PT_VERBOSE=1 timeout -k 10 300 python tools/ab.py --spp ${SPP:-256} --reps 3 "$@" 2>&1 | tee -a gpurun_out/r03b/ab.log
