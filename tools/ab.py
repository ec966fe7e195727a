#!/usr/bin/env python3
"""A/B timing of experiment builds (make variant NAME=.. DEFS=..): book-1 1080p depth 50 through every library given.
usage: tools/ab.py [--spp N] [--flags F] [--scene ID] lib1.so lib2.so ...   (paths relative to pathtracer_amd/lib/)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
args = sys.argv[1:]
spp, flags, scene_id, reps = 64, 0, 1, 3
while args and args[0].startswith("--"):
    k, v = args[0], int(args[1]); args = args[2:]
    if k == "--spp": spp = v
    elif k == "--flags": flags = v
    elif k == "--scene": scene_id = v
    elif k == "--reps": reps = v
w, h, depth = 1920, 1080, 50
import torch
for lib in args:
    pt._lib = None
    pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), lib))
    r = pt.Renderer(0)
    r.upload(pt.Scene.builtin(scene_id, w / h))
    acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    p = r.params(w, h, 0, spp, depth, 0, flags)
    r.render_device(p, acc.data_ptr(), 0); r.stats()
    ms = []
    for _ in range(reps):
        r.render_device(p, acc.data_ptr(), 0); st = r.stats(); ms.append(st["kernel_ms"])
    best = min(ms)
    print(json.dumps({"lib": lib, "Msamples/s": round(st["samples"] / best / 1e3, 1), "ms": [round(m, 2) for m in ms], "kernel": st["kernel"],
                      "nodes/seg": round(st["node_visits"] / st["segments"], 2), "tests/seg": round(st["sphere_tests"] / st["segments"], 2),
                      "checksum": float(acc.double().sum().item())}), flush=True)
    r.close()
