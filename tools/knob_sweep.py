#!/usr/bin/env python3
"""Sweep of the scheduling thresholds (environment knobs read at upload) on the bench workload at reduced spp.
usage: tools/knob_sweep.py KNOB=v1,v2,.. [KNOB=..]   e.g. PT_EXIT_LANES=4,8,12 PT_LEAF_BATCH=12,16,24"""
import itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
import torch
knobs = [a.split("=") for a in sys.argv[1:] if "=" in a]
names = [k for k, _ in knobs]
w, h, depth, spp = 1920, 1080, 50, int(os.environ.get("SWEEP_SPP", "128"))
scene = pt.Scene.builtin(int(os.environ.get("SWEEP_SCENE", "1")), w / h)
acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
for combo in itertools.product(*[v.split(",") for _, v in knobs]):
    for k, v in zip(names, combo):
        os.environ[k] = v
    r = pt.Renderer(0); r.upload(scene)
    p = r.params(w, h, 0, spp, depth, 0, 0)
    r.render_device(p, acc.data_ptr(), 0); r.stats()
    ms = []
    for _ in range(3):
        r.render_device(p, acc.data_ptr(), 0); st = r.stats(); ms.append(st["kernel_ms"])
    print(json.dumps({**dict(zip(names, combo)), "ms": round(min(ms), 3), "Msamples/s": round(st["samples"] / min(ms) / 1e3, 1),
                      "lane_util": round(st["segments"] / 64 / st["wave_iterations"], 3)}), flush=True)
    r.close()
