import os, sys, json
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pt, torch
w, h, depth, spp = 1920, 1080, 50, 512
r = pt.Renderer(0); r.upload(pt.Scene.builtin(1, w / h))
acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
p = r.params(w, h, 0, spp, depth, 0, 0)
r.render_device(p, acc.data_ptr(), 0); r.stats()
ms = []
for _ in range(3):
    r.render_device(p, acc.data_ptr(), 0); st = r.stats(); ms.append(st["kernel_ms"])
dc = r.debug_counters()
print(json.dumps({"carry_max": os.environ.get("PT_CARRY_MAX"), "ms": min(ms), "carried": dc[13], "carried_frac": dc[13] / st["samples"], "lane_util": st["segments"] / 64 / st["wave_iterations"]}))
