#!/bin/bash
# tools/profile.sh -- rocprofv3 passes over bench.py on the GPU box (one pass per run; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile.sh <tag> [bench args...]      outputs -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --cpu-spp 0 --no-brute $*"
echo "== kernel trace + stats" 
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
i=0
for PMC in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  echo "== pmc pass $i: $PMC"
  timeout -s KILL 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed: no further pass in this call"; tail -5 $OUT/pmc$i.log; exit 1; }
done
# HBM bytes per launch of this very run -> profiles/traffic.json (bench.py quotes it with the run's tag)
python3 $ROOT/tools/make_traffic.py $OUT ${PT_TRAFFIC_KEY:-scene1_1920x1080_spp512_d50} $TAG || echo "traffic.json not refreshed"
mkdir -p $ROOT/gpurun_out/profiles_new && cp $ROOT/profiles/traffic.json $ROOT/gpurun_out/profiles_new/traffic.json
echo done
