#!/usr/bin/env python3
"""Why bench.py's first timed step ran 88 ms and the others 81: every step of a fresh process on its own clock --
host time to enqueue, the kernel between its two events, wall time to completion -- with a synchronise before step SYNC_AT."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pathtracer_amd as pt
w, h, S, depth = 1920, 1080, int(os.environ.get("SPP", "512")), 50
sync_at = [int(x) for x in os.environ.get("SYNC_AT", "1").split(",")]
each = os.environ.get("SYNC_EACH", "0") == "1"
dev = torch.device("cuda", 0)
r = pt.Renderer(0); r.upload(pt.Scene.builtin(1, w / h))
accum = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
bgra = torch.empty((h, w), dtype=torch.int32, device=dev)
host = torch.empty((h, w), dtype=torch.int32, pin_memory=True)
params = r.params(w, h, 0, S, depth, seed=0)
st = torch.cuda.current_stream(dev)
prime = os.environ.get("PRIME", "")
if prime == "pinned":     # a 4-byte D2H copy into pinned memory before the first launch
    p = torch.empty(1, dtype=torch.int32, pin_memory=True); t = time.perf_counter(); p.copy_(bgra.view(-1)[:1], non_blocking=True); torch.cuda.synchronize(dev)
    print(json.dumps({"prime": prime, "ms": round((time.perf_counter() - t) * 1e3, 2)}), flush=True)
elif prime == "pageable":
    t = time.perf_counter(); bgra.view(-1)[:1].cpu()
    print(json.dumps({"prime": prime, "ms": round((time.perf_counter() - t) * 1e3, 2)}), flush=True)
elif prime == "full":     # the very copy of the step, once, with nothing in flight
    t = time.perf_counter(); host.copy_(bgra, non_blocking=True); torch.cuda.synchronize(dev)
    print(json.dumps({"prime": prime, "ms": round((time.perf_counter() - t) * 1e3, 2)}), flush=True)
elif prime == "dual":     # several D2H copies in flight at once, on streams of their own, before the first launch: every copy engine the runtime may pick later is set up here
    t = time.perf_counter()
    hosts = [torch.empty((h, w), dtype=torch.int32, pin_memory=True) for _ in range(4)]
    streams = [torch.cuda.Stream(dev) for _ in range(4)]
    for rep in range(2):
        for hb, s2 in zip(hosts, streams):
            with torch.cuda.stream(s2):
                hb.copy_(bgra, non_blocking=True)
    torch.cuda.synchronize(dev)
    print(json.dumps({"prime": prime, "ms": round((time.perf_counter() - t) * 1e3, 2)}), flush=True)
rows = []
evs = []
t_start = time.perf_counter()
for k in range(int(os.environ.get("STEPS", "8"))):
    if k in sync_at or each:
        torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record(st)
    ta = time.perf_counter()
    r.render_device(params, accum.data_ptr(), st.cuda_stream)
    t1 = time.perf_counter()
    b.record(st)
    tb = time.perf_counter()
    if os.environ.get("ZERO_COPY", "0") == "1":   # the resolve kernel packs straight into the pinned host image: no copy engine in the frame loop
        r.resolve_device(accum.data_ptr(), w * h, S, 0, host.data_ptr(), st.cuda_stream)
        tr = time.perf_counter()
    else:
        r.resolve_device(accum.data_ptr(), w * h, S, 0, bgra.data_ptr(), st.cuda_stream)
        tr = time.perf_counter()
        if os.environ.get("NO_COPY", "0") != "1":
            host.copy_(bgra, non_blocking=True)
    t2 = time.perf_counter()
    evs.append((a, b)); rows.append({"k": k, "at_ms": round((t0 - t_start) * 1e3, 1), "rec_a": round((ta - t0) * 1e3, 2), "render": round((t1 - ta) * 1e3, 2), "rec_b": round((tb - t1) * 1e3, 2),
                                     "resolve": round((tr - tb) * 1e3, 2), "copy": round((t2 - tr) * 1e3, 2)})
torch.cuda.synchronize(dev)
if os.environ.get("ZERO_COPY", "0") == "1":
    r.resolve_device(accum.data_ptr(), w * h, S, 0, bgra.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); r.resolve_device(accum.data_ptr(), w * h, S, 0, host.data_ptr(), st.cuda_stream); e1.record(st); torch.cuda.synchronize(dev)
    print(json.dumps({"zero_copy_equal_to_copied_image": bool(torch.equal(host, bgra.cpu())), "nonzero": int((host != 0).sum()), "resolve_into_host_ms": round(e0.elapsed_time(e1), 3)}), flush=True)
for row, (a, b) in zip(rows, evs):
    row["kernel_ms"] = round(a.elapsed_time(b), 2)
    if os.environ.get("ONLY_ODD", "0") != "1" or row["kernel_ms"] > 84.0 or max(row["rec_a"], row["render"], row["rec_b"], row["resolve"], row["copy"]) > 1.0 or row["k"] in sync_at:
        print(json.dumps(row), flush=True)
