#!/bin/bash
# tools/profile_c5.sh <tag> [PT_GRID_NODES kind] -- PMC passes over tools/c5_spp.py 16 (C5: 10^6 spheres, 4K, 16 spp).  outputs -> gpurun_out/prof_<tag>/
set -u
TAG=$1; KIND=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
if [ -n "$KIND" ]; then export PT_GRID_NODES=$KIND; fi
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/c5_spp.py 16"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
i=0
for PMC in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed: $PMC (no further pass in this call)"; tail -3 $OUT/pmc$i.log; exit 1; }
done
python3 $ROOT/tools/make_traffic.py $OUT scene5_n1000000_3840x2160_spp16_d50 $TAG || echo "traffic.json not refreshed"
mkdir -p $ROOT/gpurun_out/profiles_new && cp $ROOT/profiles/traffic.json $ROOT/gpurun_out/profiles_new/traffic.json
echo done
