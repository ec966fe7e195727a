import sys, time
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pt, torch
w, h = 800, 450
r = pt.Renderer(0); r.upload(pt.Scene.builtin(1, w / h))
acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0"); bgra = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
p = r.params(w, h, 0, 32, 5, 0, 0)
for fn, name in ((r.render_device, "render_device"), (r.render_device_static, "render_device_static")):
    for _ in range(5): fn(p, acc.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 200
    for _ in range(N):
        fn(p, acc.data_ptr(), s); r.resolve_device(acc.data_ptr(), w * h, 32, 0, bgra.data_ptr(), s)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(name, "host enqueue per frame %.1f us, wall per frame %.3f ms, kernel %.3f ms" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e3, r.stats()["kernel_ms"]))
