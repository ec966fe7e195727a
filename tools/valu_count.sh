#!/bin/bash
# one PMC pass: VALU instructions per launch of the bench kernel (tools/valu_count.sh <tag>)
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/valu_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --cpu-spp 0 --no-brute --steps 2 --warmup 1 > $OUT/log.txt 2>&1 || { tail -3 $OUT/log.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: "%.5g" % (sum(v) / len(v)) for k, v in agg.items()})
PY
