#!/usr/bin/env python3
"""Diagnostic: where a wave's cycles go (needs `make stamps`; never quote this build's speed, only its shares)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
scene_id = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h, spp, depth = int(os.environ.get("PH_W", "1920")), int(os.environ.get("PH_H", "1080")), int(sys.argv[2]) if len(sys.argv) > 2 else 64, 50
n5 = int(os.environ.get("PH_SPHERES", "100000"))
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for lib in ("libpt_hip_stamps.so", "libpt_hip.so"):
    pt._lib = None
    pt.load_library(os.path.join(os.path.dirname(pt.LIB_PATH), lib))
    r = pt.Renderer(0)
    r.upload(pt.Scene.builtin(scene_id, w / h, n5))
    r.render(w, h, 0, spp, depth, flags=flags)
    r.render(w, h, 0, spp, depth, flags=flags)
    st = r.stats()
    dc = r.debug_counters()
    wi = max(1, st["wave_iterations"])
    trav = {"inner_trips/iter": round(dc[9] / wi, 2), "leaf_trips/iter": round(dc[10] / wi, 2),
            "cyc_inner/iter": round(dc[11] / wi, 1), "cyc_leaf/iter": round(dc[12] / wi, 1),
            "cyc/inner_trip": round(dc[11] / max(1, dc[9]), 1), "cyc/leaf_trip": round(dc[12] / max(1, dc[10]), 1),
            "slot_iters/leaf_trip": round(dc[14] / max(1, dc[10]), 2), "exact_iters/leaf_trip": round(dc[15] / max(1, dc[10]), 2), "lane_tests/slot_iter": round(st["sphere_tests"] / max(1, dc[14]), 1)}
    print(lib, "traversal", json.dumps(trav))
    tot = sum(st["phase_cycles"]) or 1
    print(lib, "flags", flags, json.dumps({"kernel": st["kernel"], "nodes/seg": round(st["node_visits"] / max(1, st["segments"]), 2), "tests/seg": round(st["sphere_tests"] / max(1, st["segments"]), 2), "kernel_ms": round(st["kernel_ms"], 3), "Msamples/s": round(st["samples"] / st["kernel_ms"] / 1e3, 1),
          "sbar": round(st["segments"] / st["samples"], 4), "lane_util": round(st["segments"] / (64 * st["wave_iterations"]), 4),
          "groups_searched_frac": round(st["sphere_tests"] / 512 / max(1, st["wave_iterations"]) / max(1, -(-r.scene.n_spheres // 8)), 4),
          "iters_per_wave": round(st["wave_iterations"] / ((w // 8) * (h // 8)), 1),
          "share_refill": round(st["phase_cycles"][0] / tot, 4), "share_hit": round(st["phase_cycles"][1] / tot, 4),
          "share_shade": round(st["phase_cycles"][2] / tot, 4),
          "cycles_per_iter": [round(c / max(1, st["wave_iterations"]), 1) for c in st["phase_cycles"]]}))
    r.close()
