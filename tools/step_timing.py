#!/usr/bin/env python3
"""What the read-back costs inside bench.py's step: book-1 1080p 512 spp, K steps back to back, four shapes of the step."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
w, h, S, depth, K = 1920, 1080, int(os.environ.get("SPP", "512")), 50, int(os.environ.get("K", "6"))
dev = torch.device("cuda", 0)
r = pt.Renderer(0); r.upload(pt.Scene.builtin(1, w / h))
accum = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
bgra = [torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(2)]
host = torch.empty((h, w), dtype=torch.int32, pin_memory=True)
params = r.params(w, h, 0, S, depth, seed=0)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(dev)
def run(kind):
    def step(k):
        b = bgra[k & 1] if kind == "side" else bgra[0]
        r.render_device(params, accum.data_ptr(), main.cuda_stream)
        r.resolve_device(accum.data_ptr(), w * h, S, 0, b.data_ptr(), main.cuda_stream)
        if kind == "same":
            host.copy_(b, non_blocking=True)
        elif kind == "sync":
            host.copy_(b)
        elif kind == "side":
            ev = torch.cuda.Event(); ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                host.copy_(b, non_blocking=True)
    step(0); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K): step(k)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / K * 1e3
    print(json.dumps({"step": kind, "ms_per_step": round(ms, 3), "lib_kernel_ms_last": round(r.stats()["kernel_ms"], 3)}), flush=True)
for kind in ("none", "same", "sync", "side", "none", "same"):
    run(kind)
