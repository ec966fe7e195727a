#!/usr/bin/env python3
"""C5 (10^6 spheres, 3840x2160, depth 50) through the default path at several sample counts: Msamples/s against spp
(units of a few samples per pixel are all fixed cost).  usage: tools/c5_spp.py [spp ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
import torch
spps = [int(a) for a in sys.argv[1:]] or [4, 16, 64]
w, h, depth = 3840, 2160, 50
r = pt.Renderer(0)
r.upload(pt.Scene.builtin(5, w / h, 1000000))
acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
for spp in spps:
    p = r.params(w, h, 0, spp, depth, 0, 0)
    r.render_device(p, acc.data_ptr(), 0); r.stats()
    r.render_device(p, acc.data_ptr(), 0)
    st = r.stats()
    print(json.dumps({"grid_nodes": os.environ.get("PT_GRID_NODES", "default"), "spp": spp, "kernel_ms": round(st["kernel_ms"], 2), "Msamples/s": round(st["samples"] / st["kernel_ms"] / 1e3, 1),
                      "node_width": st["node_width"], "node_bytes": st["node_bytes"], "lane_util": round(st["segments"] / 64 / max(1, st["wave_iterations"]), 3),
                      "node_visits/seg": round(st["node_visits"] / max(1, st["segments"]), 1), "sphere_tests/seg": round(st["sphere_tests"] / max(1, st["segments"]), 1)}), flush=True)
