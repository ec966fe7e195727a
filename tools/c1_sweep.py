#!/usr/bin/env python3
"""C1 (book-1 800x450, 32 spp, depth 5: the reference's own default frame) against the chunk schedule's knobs.
usage: tools/c1_sweep.py KNOB=v1,v2 [KNOB=..]   (knobs are read per render call / upload)"""
import itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
import torch
knobs = [a.split("=") for a in sys.argv[1:] if "=" in a]
names = [k for k, _ in knobs]
w, h, depth, spp = int(os.environ.get("C1_W", "800")), int(os.environ.get("C1_H", "450")), int(os.environ.get("C1_DEPTH", "5")), int(os.environ.get("C1_SPP", "32"))
scene = pt.Scene.builtin(1, w / h)
acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
for combo in itertools.product(*[v.split(",") for _, v in knobs]):
    for k, v in zip(names, combo):
        os.environ[k] = v
    r = pt.Renderer(0); r.upload(scene)
    p = r.params(w, h, 0, spp, depth, 0, 0)
    r.render_device(p, acc.data_ptr(), 0); r.stats()
    ms = []
    for _ in range(8):
        r.render_device(p, acc.data_ptr(), 0); st = r.stats(); ms.append(st["kernel_ms"])
    ms.sort()
    print(json.dumps({**dict(zip(names, combo)), "ms_min": round(ms[0], 4), "ms_median": round(ms[len(ms) // 2], 4), "Msamples/s": round(st["samples"] / ms[0] / 1e3, 1),
                      "lane_util": round(st["segments"] / 64 / st["wave_iterations"], 3)}), flush=True)
    r.close()
