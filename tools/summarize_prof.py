#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (rocprofv3 csv) into one text file for profiles/."""
import csv, glob, os, sys, collections
d = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"# {os.path.relpath(f, d)}")
    out.append(open(f).read().strip())
for f in sorted(glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    out.append(f"# {os.path.relpath(f, d)}  (sum over dispatches; n = dispatches)")
    for k, cs in agg.items():
        for c, v in cs.items():
            out.append(f"{k:60s} {c:28s} sum={v:.6g} n={cnt[(k, c)]} per_dispatch={v / cnt[(k, c)]:.6g}")
print("\n".join(out))
