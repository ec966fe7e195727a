#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path tracer (contract: see the task's bench section).

Metric   : Msamples/s, whole job (BASELINE.json: "Msamples/s (whole node) book-1 scene 1080p @1/2/4/8 GPUs").
Workload : BASELINE.json configs[2] geometry -- init_test_scene_1 (book-1 final: 259 spheres + 1 plane), 1920x1080,
           depth 50, 512 spp PER GPU (4096 spp split across 8 GPUs) -- so per-GPU work is fixed as N grows (weak
           scaling): rank r renders global sample indices [r*512, (r+1)*512) of every pixel.
Step     : one complete frame: render kernel -> (N>1) RCCL all-reduce of the float accumulation buffer -> device
           resolve (clamp(mean), gamma, BGRA pack) -> read-back of the packed image to pinned host memory on rank 0
           (the reference's timed span, main.cpp:412-481, ends with the pixels in host memory).  Scene and
           working buffers are resident in HBM before the timed region.
One rank per GPU.  N>1: launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env), or --
with no WORLD_SIZE in the env -- bench.py starts the N ranks itself as child processes BEFORE anything touches a GPU
(the parent imports neither torch nor the library; it exits non-zero if any rank does).
--spp-total T: strong scaling -- T samples per pixel split over the ranks (C3: --spp-total 4096).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (= 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz): packed v_pk_fma_f32
PEAK_FP32_UNPACKED = 78.6  # ... and with one FP32 FMA per lane per issue (report against both)
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def flops_per_sample(sbar, n_spheres, n_planes, moving, box_tests_per_seg=None, leaf_tests_per_seg=None):
    """Algorithmic FLOPs per sample, SURVEY.md 8(d): ray-gen 90 + per segment (14/plane + sphere query) + 75 per
    hit-segment shading + 8 sky terminal.  Sphere query: brute force = 25|31 per sphere of the scene; BVH kernels =
    32 per box test + 25|31 per leaf sphere test, both taken from the kernel's own device counters."""
    f_sph = 31.0 if moving else 25.0
    if box_tests_per_seg is None:
        query = f_sph * n_spheres
    else:
        query = 32.0 * box_tests_per_seg + f_sph * leaf_tests_per_seg
    return 90.0 + sbar * (14.0 * n_planes + query) + (sbar - 1.0) * 75.0 + 8.0


def cpu_baseline(scene_id, w, h, depth, spp, threads, n_spheres=0):
    """The reference's own renderer (compiled into oracle/_ref) -- or, where that library is absent, our CPU
    restatement -- timed on the host cores on a bounded sample of the same workload (same scene, resolution and
    depth; fewer spp -- Msamples/s does not depend on spp)."""
    from oracle import pyoracle as po
    import pathtracer_amd as pt
    from tests._util import to_scene_arrays
    sa = to_scene_arrays(pt.Scene.builtin(scene_id, w / h, n_spheres))
    o3 = None
    if not po.Ref.available():
        print("bench.py: oracle/_ref/libptref.so (the compiled reference) is ABSENT -- it is built from /root/reference by "
              "`make -C oracle`, which this tree has not seen; cpu_baseline falls back to kind \"port\" (our CPU restatement)",
              file=sys.stderr)
    if po.Ref.available() and sa.n_spheres <= 4192:   # the reference's World holds 4192 objects (render_world.h:85)
        R = po.Ref()
        hs = R.scene_from_arrays(sa)
        R.build_bvh(hs, 0)
        secs, _ = R.render(hs, w, h, spp, depth, threads)
        R.scene_free(hs)
        kind = "reference"
        # the courtesy figure SURVEY.md 8(d) asks for: the same reference sources at -O3 -march=native (the reference's
        # own flag is /O2, build.bat:13), when oracle/Makefile built that flavour for THIS machine's CPU
        if po.Ref.available(flavour="o3"):
            try:
                R3 = po.Ref(flavour="o3")
                h3 = R3.scene_from_arrays(sa)
                R3.build_bvh(h3, 0)
                s3, _ = R3.render(h3, w, h, spp, depth, threads)
                R3.scene_free(h3)
                o3 = {"value": round(w * h * spp / s3 / 1e6, 4), "flags": "-O3 -march=x86-64-v3 (stands in for -march=native: built in the container, run on this host)"}
            except Exception as e:  # noqa: BLE001 - e.g. built for another CPU
                o3 = {"value": None, "error": str(e)[:80]}
    else:
        O = po.Oracle()
        hs = O.scene(sa)
        secs, _, _ = O.time_render_philox(hs, w, h, 0, spp, depth, 0, threads)
        O.scene_free(hs)
        kind = "port"
    return {"value": round(w * h * spp / secs / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": kind,
            "sample": f"scene {scene_id} {w}x{h} depth {depth}, {spp} spp ({w * h * spp / 1e6:.1f} Msamples, {secs:.1f} s wall)",
            "flags": "-O2 (the reference's /O2, build.bat:13)" if kind == "reference" else "-O2", "o3_native": o3}


def host_threads():
    """Threads for the CPU baseline = the host cores this process may use: its affinity mask, further capped by the
    cgroup CPU quota when one is set (a GPU box hands one GPU's job a share of the machine; threads beyond the quota
    only get throttled).  Returns (threads, {"os_cpu_count", "affinity", "cgroup_quota"}) so the line states all three."""
    n_aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    n = n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5)))
    return n, {"os_cpu_count": os.cpu_count(), "affinity": n_aff, "cgroup_quota": quota}


def launch_ranks(n, argv):
    """N > 1 with no launcher above us: start the N ranks as fresh child processes, each with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in its environment (what torch.distributed.run would set).  Runs BEFORE anything in this process
    has touched a GPU -- this parent imports neither torch nor libpt_hip.so -- and replaces nothing: it waits for its
    children and exits non-zero if any of them does (the others are then stopped)."""
    assert "torch" not in sys.modules and "pathtracer_amd" not in sys.modules
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = dict(enumerate(procs))
    while pending:
        for r, pr in list(pending.items()):
            code = pr.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr)
                for other in pending.values():
                    other.terminate()
        time.sleep(0.05)
    return rc


def pick_device(local_rank, n_visible):
    """The GPU of a rank: LOCAL_RANK when the process sees the node's GPUs, LOCAL_RANK modulo what it does see when a
    launcher masked the devices per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES: each rank then sees ONE device, 0)."""
    return local_rank % n_visible if n_visible > 0 else -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--spheres", type=int, default=0, help="sphere count of scene 5 (C5: 1000000)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-gpu", type=int, default=512)
    ap.add_argument("--spp-total", type=int, default=0, help="strong scaling: this many samples per pixel split over the ranks (C3: 4096); 0 = weak (--spp-per-gpu each)")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-spp", type=int, default=16, help="spp of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-brute", action="store_true", help="skip the extra launch of the brute-force kernel (profiling runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))   # (no GPU call has been made in this process)

    if os.environ.get("PT_BENCH_WATCHDOG"):   # rehearsals: a rank that hangs says where and ends
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PT_BENCH_WATCHDOG"]), exit=True)

    import torch
    import torch.distributed as dist
    import pathtracer_amd as pt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count wins nothing here, give both the same N", file=sys.stderr)
        sys.exit(2)
    n_visible = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    dev_index = pick_device(local_rank, n_visible)
    if dev_index < 0 or not torch.cuda.is_available() or pt.device_count() <= dev_index:
        print("bench.py: no HIP device visible -- the product path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if world > n_visible and n_visible > 1 and rank == 0:
        print(f"bench.py: {world} ranks share {n_visible} visible GPUs", file=sys.stderr)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # PT_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, all-reduce, barrier) with a single rank too -- a
    # rehearsal of the N > 1 plumbing on a one-GPU box; the numbers of such a run are N = 1 numbers.
    use_dist = world > 1 or os.environ.get("PT_BENCH_FORCE_DIST") == "1"
    stdout_fd = None
    if use_dist:
        # the communicator libraries print banners on stdout (RCCL's version block, gloo's "connected to N peer ranks"):
        # stdout is for the ONE JSON line, so everything else this process prints goes to stderr until then
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # nccl == RCCL on ROCm.  PT_BENCH_BACKEND=gloo: rehearsal of the N > 1 plumbing (launcher, slices, reduce, max over
        # ranks) on a box with fewer GPUs than ranks -- RCCL refuses two ranks on one device; such a line says so.
        backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # (the rehearsal backend reduces through host memory: gloo is a CPU transport)
    staged = use_dist and os.environ.get("PT_BENCH_BACKEND", "nccl") != "nccl"
    red_dev = torch.device("cpu") if staged else dev

    w, h, depth = args.width, args.height, args.depth
    # this rank's slice of the global sample indices: weak = S each, strong = its part of --spp-total
    if args.spp_total > 0:
        spp_all = args.spp_total
        s_begin, s_end = pt.spp_slice(0, spp_all, world, rank)
    else:
        spp_all = args.spp_per_gpu * world
        s_begin, s_end = rank * args.spp_per_gpu, (rank + 1) * args.spp_per_gpu
    S = s_end - s_begin
    if S <= 0:
        print(f"bench.py: rank {rank} has no samples ({spp_all} spp over {world} ranks)", file=sys.stderr)
        sys.exit(2)
    scene = pt.Scene.builtin(args.scene, w / h, args.spheres)
    r = pt.Renderer(dev_index)
    r.upload(scene)
    accum = torch.empty((h, w, 4), dtype=torch.float32, device=dev)
    bgra = torch.empty((h, w), dtype=torch.int32, device=dev)
    bgra_host = torch.empty((h, w), dtype=torch.int32, pin_memory=True) if rank == 0 else None
    params = r.params(w, h, s_begin, s_end, depth, seed=args.seed)
    stream = torch.cuda.current_stream(dev)

    def step(ev=None):
        if ev:
            ev[0].record(stream)
        r.render_device(params, accum.data_ptr(), stream.cuda_stream)
        if ev:
            ev[1].record(stream)
        if staged:
            tmp = accum.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
            accum.copy_(tmp)
        elif use_dist:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)  # float accumulation buffer over xGMI, before tonemapping
        if rank == 0:
            # resolve + BGRA pack written by the kernel STRAIGHT into the pinned host image: 4 B/pixel over PCIe inside the
            # timed step, 0.16 ms at 1080p.  (Through a device image and the runtime's copy engine it costs
            # the same 0.16 ms, but the runtime sets that path up lazily -- 9 ms host-side calls that stall the kernel in
            # flight, the first timed launch read 88 ms instead of 81 in half of all runs: profiles/README.md.)
            r.resolve_device(accum.data_ptr(), w * h, spp_all, 0, bgra_host.data_ptr(), stream.cuda_stream)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))   # the timed step exactly, events included
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms_steps = [a.elapsed_time(b) for a, b in events]
    kernel_ms = sum(kernel_ms_steps) / max(1, args.steps)
    st = r.stats()  # last launch: samples, segments (S-bar), library's own HIP-event time

    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        seg = torch.tensor([float(st["segments"]), float(st["samples"]), kernel_ms, float(st["node_visits"]), float(st["sphere_tests"])],
                           dtype=torch.float64, device=red_dev)
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        segs, smps, kernel_ms, nvis, stest = float(seg[0]), float(seg[1]), float(seg[2]) / world, float(seg[3]), float(seg[4])
    else:
        segs, smps, nvis, stest = float(st["segments"]), float(st["samples"]), float(st["node_visits"]), float(st["sphere_tests"])

    host_img = bgra_host if rank == 0 else None
    if rank == 0:
        # what the read-back costs, by itself: the packed image (in the step) and the float sums it replaces (8f rank 1)
        def timed_copy(dst, src):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t) * 1e3
        accum_host = torch.empty((h, w, 4), dtype=torch.float32, pin_memory=True)
        bgra_tmp = torch.empty((h, w), dtype=torch.int32, pin_memory=True)

        def timed_resolve(dst_ptr):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            r.resolve_device(accum.data_ptr(), w * h, spp_all, 0, dst_ptr, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t) * 1e3
        readback_ms = {"resolve_into_pinned_host_4B_per_pixel (the step's)": round(min(timed_resolve(bgra_tmp.data_ptr()) for _ in range(3)), 3),
                       "resolve_into_device_image": round(min(timed_resolve(bgra.data_ptr()) for _ in range(3)), 3),
                       "copy_engine_bgra_4B_per_pixel": round(min(timed_copy(bgra_tmp, bgra) for _ in range(3)), 3),
                       "copy_engine_f32_sums_16B_per_pixel": round(min(timed_copy(accum_host, accum) for _ in range(3)), 3)}
        # (the image the kernel wrote into pinned host memory during the last step against the same frame resolved on the
        # device and copied: reported on the line, loudly on stderr if it ever differs -- never at the cost of the line)
        readback_ms["pinned_image_equals_copied_image"] = bool(torch.equal(bgra_tmp, bgra_host))
        if not readback_ms["pinned_image_equals_copied_image"]:
            print("bench.py: the image resolved straight into pinned host memory DIFFERS from the copied one", file=sys.stderr)
        del accum_host, bgra_tmp
        samples_per_step = w * h * spp_all
        value = samples_per_step * args.steps / elapsed / 1e6
        sbar = segs / smps
        moving = scene.vel is not None
        kernel_name = {0: "k_render<brute force, scene in LDS>", 1: "k_render<BVH in HBM>", 2: "k_render<strict search>", 3: "k_render<BVH in LDS>", 5: "k_render<straight loop over leaf blocks, no tree>"}[st["kernel"]]
        is_bvh = st["kernel"] in (1, 3)
        # an inner-node visit tests every child box it holds (2 or 4 x 32 FLOP); leaf tests are per sphere
        f_sample = flops_per_sample(sbar, scene.n_spheres, scene.n_planes, moving, float(st["node_width"] or 2) * nvis / segs, stest / segs) if is_bvh else \
            flops_per_sample(sbar, scene.n_spheres, scene.n_planes, moving)
        # roofline of the dominant kernel (k_render), per launch on one GPU (mean over the ranks' launches)
        achieved = (samples_per_step / world) * f_sample / (kernel_ms * 1e-3) / 1e12
        # HBM bytes per launch come from PMC passes (rocprofv3 cannot run inside this process): tools/profile.sh runs
        # THIS command under the counters and writes profiles/traffic.json in the same run that produces the kept
        # summary; the line says which run that was.  No entry for this workload -> null.
        traffic, traffic_run, retired = None, None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                t = json.load(open(tf)).get(f"scene{args.scene}{'_n' + str(args.spheres) if args.spheres else ''}_{w}x{h}_spp{S}_d{depth}", {})
                traffic, traffic_run = t.get("hbm_bytes_per_launch"), t.get("run")
                # FP32 work the SIMDs RETIRED in that profiled launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 x 64 lanes x the
                # launch's lane activity, FMA = 2), over the profiled kernel time: the counters' own fraction of the peak
                if t.get("retired_fp32_flop_per_launch") and t.get("kernel_ns"):
                    rt = t["retired_fp32_flop_per_launch"] / (t["kernel_ns"] * 1e-9) / 1e12
                    retired = {"frac": round(rt / PEAK_FP32_TFLOPS, 4), "achieved": round(rt, 2), "lane_activity": t.get("lane_activity"),
                               "valu_insts_per_launch": t.get("valu_insts_per_launch"), "run": t.get("run")}
            except Exception:  # noqa: BLE001
                traffic = None
        # The formulation SURVEY.md 8(d) prices (brute force over the LDS-resident scene, packed FP32), MEASURED: one
        # launch of that kernel on the same workload, outside the timed region.
        brute = None
        if world == 1 and is_bvh and not args.no_brute:
            try:
                pb = r.params(w, h, 0, S, depth, seed=args.seed, flags=pt.FLAG_BRUTE_FORCE)
                r.render_device(pb, accum.data_ptr(), stream.cuda_stream)
                sb = r.stats()
                fb = flops_per_sample(sb["segments"] / sb["samples"], scene.n_spheres, scene.n_planes, moving)
                tfl = sb["samples"] * fb / (sb["kernel_ms"] * 1e-3) / 1e12
                brute = {"kernel": "k_render<brute force, scene in LDS> (PT_FLAG_BRUTE_FORCE), one launch outside the timed region",
                         "kernel_ms": round(sb["kernel_ms"], 3), "Msamples/s": round(sb["samples"] / sb["kernel_ms"] / 1e3, 1),
                         "flop_per_sample": round(fb, 1), "achieved": round(tfl, 2), "frac": round(tfl / PEAK_FP32_TFLOPS, 4)}
            except pt.PtError:
                brute = None   # the scene has no brute-force image in LDS
        roof = {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "frac_unpacked": round(achieved / PEAK_FP32_UNPACKED, 4),
                "retired": retired, "traffic": traffic, "traffic_run": traffic_run,
                "kernel": kernel_name, "flop_per_sample": round(f_sample, 1),
                "box_tests_per_segment": round((st["node_width"] or 2) * nvis / segs, 2) if is_bvh else None,
                "leaf_sphere_tests_per_segment": round(stest / segs, 2) if is_bvh else None,
                "brute_force_kernel": brute,
                "note": "FP32 vector (VALU) roofline, not MFMA (BASELINE.json north_star). Algorithmic FLOPs per SURVEY.md 8(d): "
                        "for the BVH kernels the sphere term is 32/box test + 25|31/leaf test from device counters, so frac is the "
                        "share of peak spent on algorithmically necessary FP32 work; brute_force_kernel is the measured figure of "
                        "the brute-force formulation 8(d) prices (its own FLOPs over its own time)"}
        if st["kernel"] == 1 and scene.n_spheres >= 32768:
            # Big trees in HBM (C5).  SURVEY.md 8(d) prices them by algorithmic bytes -- per segment node bytes x nodes visited
            # + 32 B x leaf spheres tested, per pixel one 16-byte sum -- but the L2 and the Infinity Cache serve nearly all of
            # them (the counters see ~1/30 of that at the HBM interface) and the kernel is bound by VALU issue: the fraction
            # on the line is the FP32 one, the HBM figure is the COUNTER traffic over the kernel time, the algorithmic bytes
            # ride along as a side field.
            nb = float(st["node_bytes"])
            bytes_launch = segs / world * (nb * nvis / segs + 32.0 * stest / segs) + 16.0 * w * h
            roof["kernel"] = kernel_name + f" ({st['node_width']}-wide nodes of {st['node_bytes']} B)"
            roof["algorithmic_bytes_per_launch"] = int(bytes_launch)
            roof["node_visits_per_segment"] = round(nvis / segs, 2)
            roof["hbm"] = None if traffic is None else {"achieved": round(traffic / (kernel_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                        "frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "source": "PMC traffic / live kernel time"}
            roof["note"] = ("VALU-issue bound (profiles/README.md: lane activity ~0.55, L2 hit ~95 %): frac is the FP32 one (32 FLOP per child box tested "
                            "+ 25|31 per leaf test, device counters); hbm = counter traffic, algorithmic_bytes_per_launch = SURVEY.md 8(d)'s byte figure "
                            "(served by L2 / Infinity Cache, so it is no HBM roofline numerator)")
        assert 0.0 <= roof["frac"] <= 1.0, roof
        name = "book-1 scene 1080p" if (args.scene, w, h) == (1, 1920, 1080) else f"scene {args.scene} {w}x{h}"
        out = {
            "metric": f"Msamples/s (whole node) {name}", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.spp_total > 0 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'init_test_scene_' + str(args.scene) if args.scene <= 3 else 'build-defined scene ' + str(args.scene)} ({'book-1 final, ' if args.scene == 1 else ''}{scene.n_spheres} spheres + {scene.n_planes} plane) "
                                   f"{w}x{h}, depth {depth}, " + (f"{spp_all} spp total split over {world} GPU(s)" if args.spp_total > 0 else f"{S} spp per GPU = {spp_all} spp total"),
                       "global_samples_per_step": samples_per_step, "parallelism": f"spp-split x{world} + RCCL all-reduce(sum) of the f32 accumulation buffer" + (f" [REHEARSAL: backend {os.environ['PT_BENCH_BACKEND']}, {n_visible} GPU(s) visible]" if use_dist and os.environ.get("PT_BENCH_BACKEND", "nccl") != "nccl" else ""),
                       "segments_per_sample": round(sbar, 4), "lane_utilisation": round(st["segments"] / (64.0 * max(1, st["wave_iterations"])), 4), "kernel_ms": round(kernel_ms, 3), "kernel_ms_steps": [round(x, 2) for x in kernel_ms_steps[:16]], "lib_kernel_ms_last": round(st["kernel_ms"], 3),
                       "step": "render -> " + ("all-reduce(sum, f32) -> " if use_dist else "") + "resolve + BGRA pack written straight into the pinned host image (4 B/pixel over PCIe), all inside the timed region",
                       "readback_ms": readback_ms},
            "roofline": roof,
        }
        if world == 1 and args.cpu_spp > 0:
            threads, host = host_threads()
            try:
                out["cpu_baseline"] = cpu_baseline(args.scene, w, h, depth, args.cpu_spp, threads, args.spheres)
                out["cpu_baseline"]["host"] = host
                out["config"]["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
            except Exception as e:  # noqa: BLE001 - the baseline is a reported side figure: its failure must not cost the GPU line
                print(f"bench.py: cpu_baseline failed: {e!r}", file=sys.stderr)
                out["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": threads, "kind": "failed", "sample": str(e)[:200], "host": host}
        assert host_img is not None and host_img.shape == (h, w)
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
            os.close(stdout_fd)
            stdout_fd = None
        print(json.dumps(out), flush=True)
        if use_dist:
            os.dup2(2, 1)   # (teardown chatter stays off stdout too)
    r.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
