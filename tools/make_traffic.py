#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile.sh run: HBM bytes per launch of k_render from the PMC counters -- and the
FP32 work the SIMDs retired in that launch (bench.py's roofline.retired) -- 
corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reads HALF the bytes
of a wide coalesced read (x2); WRITE_SIZE is exact for 16-B-per-lane stores (our only stores)."""
import csv, glob, json, os, sys
d, key = sys.argv[1], sys.argv[2]
run = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.path.normpath(d))
vals = {}
WANT = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
        "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"] and r["Counter_Name"] in WANT:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
mean = lambda k: sum(vals[k]) / len(vals[k])
extra = {}
try:
    # wave-level instruction counts x 64 lanes x the launch's lane activity (thread-cycles / (active VALU cycles x 64)); FMA = 2
    act = mean("SQ_THREAD_CYCLES_VALU") / (mean("SQ_ACTIVE_INST_VALU") * 64.0)
    flop = (2.0 * mean("SQ_INSTS_VALU_FMA_F32") + mean("SQ_INSTS_VALU_ADD_F32") + mean("SQ_INSTS_VALU_MUL_F32") + mean("SQ_INSTS_VALU_TRANS_F32")) * 64.0 * act
    extra = {"retired_fp32_flop_per_launch": flop, "lane_activity": round(act, 4), "valu_insts_per_launch": mean("SQ_INSTS_VALU")}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render" in r["Name"]:
                extra["kernel_ns"] = float(r["AverageNs"])
except (KeyError, ZeroDivisionError):
    pass
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]); write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
out = {key: {"hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
             "correction": "2*FETCH_SIZE + WRITE_SIZE, KB->B (MI355X_MICROARCH.md, HBM)", "run": run, **extra}}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
old = json.load(open(path)) if os.path.exists(path) else {}
old.update(out)
json.dump(old, open(path, "w"), indent=1)
print(json.dumps(out))
