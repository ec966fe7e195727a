#!/usr/bin/env python3
"""A/B of experiment builds on C5 (10^6 spheres, 3840x2160, depth 50).  usage: tools/ab_c5.py [--spp N] lib1.so lib2.so ... (relative to pathtracer_amd/lib/)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
import torch
args = sys.argv[1:]
spp = 8
if args and args[0] == "--spp":
    spp = int(args[1]); args = args[2:]
w, h, depth = 3840, 2160, 50
scene = pt.Scene.builtin(5, w / h, 1000000)
for lib in args:
    pt._lib = None
    pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), lib))
    r = pt.Renderer(0)
    r.upload(scene)
    acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    p = r.params(w, h, 0, spp, depth, 0, 0)
    r.render_device(p, acc.data_ptr(), 0); r.stats()
    ms = []
    for _ in range(3):
        r.render_device(p, acc.data_ptr(), 0); st = r.stats(); ms.append(st["kernel_ms"])
    print(json.dumps({"lib": lib, "Msamples/s": round(st["samples"] / min(ms) / 1e3, 1), "ms": [round(m, 2) for m in ms], "node_width": st["node_width"],
                      "nodes/seg": round(st["node_visits"] / st["segments"], 2), "tests/seg": round(st["sphere_tests"] / st["segments"], 2),
                      "lane_util": round(st["segments"] / 64 / max(1, st["wave_iterations"]), 3), "checksum": float(acc.double().sum().item())}), flush=True)
    r.close()
