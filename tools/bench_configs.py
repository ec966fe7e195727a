#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configurations on ONE GPU at bounded spp (Msamples/s does not depend on spp),
each through the kernel family it selects -- and, where both apply, through both.  Prints one JSON line per case."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt

CASES = [  # name, scene id, n (id 5), w, h, spp, depth, flags
    ("C1 book-1 800x450 32spp d5", 1, 0, 800, 450, 32, 5, 0),
    ("C2 materials+DOF 1080p", 4, 0, 1920, 1080, 64, 50, 0),
    ("C3 book-1 1080p d50 (default: tree in LDS)", 1, 0, 1920, 1080, 64, 50, 0),
    ("C3 book-1 1080p d50 (brute force in LDS)", 1, 0, 1920, 1080, 64, 50, pt.FLAG_BRUTE_FORCE),
    ("C3 book-1 1080p d50 (tree in HBM)", 1, 0, 1920, 1080, 64, 50, pt.FLAG_FORCE_BVH),
    ("C4 motion blur 4K", 3, 0, 3840, 2160, 32, 50, 0),
    ("scene 2 (2400 spheres) 1080p (BVH)", 2, 0, 1920, 1080, 8, 50, 0),
    ("C5 1e6 spheres 4K (BVH)", 5, 1000000, 3840, 2160, 4, 50, 0),
]
only = sys.argv[1:] 
r = pt.Renderer(0)
for name, sid, n, w, h, spp, depth, flags in CASES:
    if only and not any(o in name for o in only):
        continue
    t0 = time.time()
    scene = pt.Scene.builtin(sid, w / h, n)
    t1 = time.time()
    r.upload(scene)
    t2 = time.time()
    import torch
    acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    p = r.params(w, h, 0, spp, depth, 0, flags)
    r.render_device(p, acc.data_ptr(), 0); r.stats()
    r.render_device(p, acc.data_ptr(), 0)
    st = r.stats()
    print(json.dumps({"case": name, "spheres": scene.n_spheres, "kernel": {0: "lds-brute", 1: "bvh-hbm", 2: "lds-strict", 3: "bvh-lds", 4: "reference-tree", 5: "few-spheres loop"}[st["kernel"]], "samples": st["samples"],
                      "kernel_ms": round(st["kernel_ms"], 2), "Msamples/s": round(st["samples"] / st["kernel_ms"] / 1e3, 1),
                      "sbar": round(st["segments"] / st["samples"], 3), "lane_util": round(st["segments"] / 64 / max(1, st["wave_iterations"]), 3),
                      "node_visits/seg": round(st["node_visits"] / max(1, st["segments"]), 1), "sphere_tests/seg": round(st["sphere_tests"] / max(1, st["segments"]), 1),
                      "scene_build_s": round(t1 - t0, 2), "upload_s(bvh build)": round(t2 - t1, 2)}), flush=True)

# the same scene on a node of ONE GPU through pt_node_upload_scene: one host-side prepare (tree built once), one commit
if not only or any("node" in o for o in only):
    scene = pt.Scene.builtin(5, 3840 / 2160, 1000000)
    with pt.Node(1) as node:
        b0 = pt.tree_build_count(); t0 = time.time()
        node.upload(scene)
        t1 = time.time()
        print(json.dumps({"case": "C5 upload through pt.Node(1) (pt_scene_prepare once + pt_hip_commit_scene)", "spheres": scene.n_spheres,
                          "upload_s": round(t1 - t0, 3), "tree_builds": pt.tree_build_count() - b0}), flush=True)
