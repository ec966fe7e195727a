#!/usr/bin/env python3
"""tools/report.py -- the report SURVEY.md 8(d) asks for, in one run on the GPU box: throughput of the BASELINE.json
configurations with their roofline figures, the CPU reference beside it, and the parity gates.

  gate 1  per-ray / per-function: records compared and records equal, GPU (through the C-ABI) against the reference's
          recorded outputs (tests/golden, generated from the compiled reference) on identical inputs
  gate 2  converged image against the reference renderer (PSNR, mean relative error, with the reference-vs-reference
          noise floor), in the default closest-hit mode and in PT_FLAG_REFERENCE_QUERY mode
  gate 3  one sample range rendered in one piece against the sum of 8 slices (what 8 GPUs would all-reduce)

Prints one JSON object.  The oracle is used here as the checker only (this tool is test infrastructure, like tests/)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests._util import bits, to_pt_scene, to_scene_arrays, psnr, mean_rel_error  # noqa: E402
import bench  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PEAK = bench.PEAK_FP32_TFLOPS


def eq_rows(a, b):
    a, b = bits(a), bits(b)
    return int((a.reshape(len(a), -1) == b.reshape(len(b), -1)).all(axis=1).sum())


def gate1(r, O):
    out = {}
    z = np.load(os.path.join(GOLD, "prims.npz"))
    n = len(z["plane_rays"])
    ts, _ = r.probe_primitives(z["sphere_rays"], z["spheres"], np.tile(z["planes"], (4, 1)))
    _, tp = r.probe_primitives(z["plane_rays"], z["spheres"][:n], z["planes"])
    out["ray_sphere_t (geometry.cpp:39-68)"] = [eq_rows(ts, z["sphere_t"]), len(ts)]
    out["ray_plane_t (geometry.cpp:70-81)"] = [eq_rows(tp, z["plane_t"]), len(tp)]
    zc, rngz, sc = np.load(os.path.join(GOLD, "cameras.npz")), np.load(os.path.join(GOLD, "rng.npz")), np.load(os.path.join(GOLD, "scenes.npz"))
    ok = tot = 0
    for sid in (1, 3):
        sa = po.SceneArrays.from_npz(sc, f"s{sid}_")
        r.upload(to_pt_scene(sa))
        _, _, ofs = O.get_rays(zc[f"cam{sid}"], 3, zc[f"uv{sid}"])
        rays = r.probe_get_rays(zc[f"uv{sid}"], rngz["f32"][3:3 + int(ofs[-1])], ofs)
        ok += eq_rows(rays, zc[f"rays{sid}"]); tot += len(rays)
    out["get_ray (camera.cpp:57-79)"] = [ok, tot]
    okq = totq = okp = totp = 0
    worst = 0.0
    for name in ("scene1", "scene3", "random_moving", "materials"):
        z = np.load(os.path.join(GOLD, f"paths_{name}.npz"))
        sa = po.SceneArrays.from_npz(z)
        r.upload(to_pt_scene(sa))
        r.build_reference_tree(0)
        h = O.scene(sa)
        O.scene_set_bvh(h, z["boxes"], z["links"])
        rays, times = z["rays"], z["times"]
        # the BVH query (main.cpp:53-91) in reference-query mode against the reference's own (t, object)
        t, prim, _, _ = r.probe_hits(rays, times, pt.FLAG_REFERENCE_QUERY)
        to, po_, _, _ = O.closest_hits(h, rays, times, po.QUERY_BVH_QUIRK)
        okq += int(((prim == po_) & (bits(t) == bits(to))).sum()); totq += len(rays)
        # cast_ray (main.cpp:171-307), depth 50, on the reference's own random streams
        ref = z["rgba_d50"]
        _, _, _, o32, o64, _ = O.cast_rays_mt(h, rays, times, 50, 11, 3, po.QUERY_BVH_QUIRK, po.PRODUCT_RECURSIVE)
        t32, t64 = O.mt_f32(11, int(o32[-1]) + 64), O.mt_f64(3, int(o64[-1]) + 8)
        c, _ = r.probe_paths(rays, times, 50, t32, o32, t64, o64, pt.FLAG_REFERENCE_QUERY)
        rel = np.abs(c.astype(np.float64) - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1e-6)
        okp += int((rel.max(axis=1) <= 1e-5).sum()); totp += len(rays); worst = max(worst, float(rel.max()))
        O.scene_free(h)
    out["closest hit incl. planes, reference-query mode (main.cpp:180-216, 53-91)"] = [okq, totq]
    out["cast_ray depth 50 within 1e-5 of the reference, reference-query mode (main.cpp:171-307)"] = [okp, totp]
    out["cast_ray worst relative difference"] = worst
    return out


def gate2(r):
    z = np.load(os.path.join(GOLD, "converged_images.npz"))
    out = {}
    for key, sid in (("scene1", 1), ("scene3", 3)):
        ref, w, h, spp, depth = z[key + "_img"], int(z[key + "_w"]), int(z[key + "_h"]), int(z[key + "_spp"]), int(z[key + "_depth"])
        r.upload(pt.Scene.builtin(sid, w / h))
        r.build_reference_tree(0)
        row = {"size": [w, h], "spp": spp, "depth": depth, "reference_vs_reference_floor": {"psnr_db": round(float(z[key + "_floor_psnr"]), 2), "mean_rel_err": round(float(z[key + "_floor_mre"]), 5)}}
        for mode, flags in (("default (closest hit, D1)", 0), ("PT_FLAG_REFERENCE_QUERY", pt.FLAG_REFERENCE_QUERY)):
            img = pt.resolve(r.render(w, h, 0, spp, depth, seed=123, flags=flags), spp).reshape(h, w, 4)
            row[mode] = {"psnr_db": round(psnr(img[..., :3], ref[..., :3]), 2), "mean_rel_err": round(mean_rel_error(img, ref), 5)}
        out[key] = row
    return out


def gate3(r):
    w, h, spp, depth, parts = 1920, 1080, 64, 50, 8
    r.upload(pt.Scene.builtin(1, w / h))
    whole = r.render(w, h, 0, spp, depth, seed=0).astype(np.float64)
    total = np.zeros_like(whole)
    for g in range(parts):
        a, b = pt.spp_slice(0, spp, parts, g)
        total += r.render(w, h, a, b, depth, seed=0)
    d = np.abs(whole - total)
    return {"workload": f"book-1 {w}x{h} depth {depth}, {spp} spp in one piece vs {parts} slices summed", "max_abs_diff": float(d.max()),
            "max_rel_diff_of_pixel_sums": float((d / np.maximum(np.abs(whole), 1.0)).max()), "alpha_sum_exact": bool((total[..., 3] == spp).all())}


def throughput(r):
    import torch
    cases = [("C1 book-1 800x450 32 spp depth 5", 1, 0, 800, 450, 32, 5), ("C2 materials + DOF 1080p", 4, 0, 1920, 1080, 256, 50),
             ("C3 book-1 1080p (per-GPU slice of 4096 spp)", 1, 0, 1920, 1080, 512, 50), ("C4 motion blur 4K", 3, 0, 3840, 2160, 128, 50),
             ("C5 1e6 spheres 4K", 5, 1000000, 3840, 2160, 16, 50)]
    rows = []
    for name, sid, n, w, h, spp, depth in cases:
        scene = pt.Scene.builtin(sid, w / h, n)
        r.upload(scene)
        acc = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
        p = r.params(w, h, 0, spp, depth, 0, 0)
        r.render_device(p, acc.data_ptr(), 0); r.stats()
        r.render_device(p, acc.data_ptr(), 0)
        st = r.stats()
        sbar = st["segments"] / st["samples"]
        f = bench.flops_per_sample(sbar, scene.n_spheres, scene.n_planes, scene.vel is not None,
                                   float(st["node_width"] or 2) * st["node_visits"] / st["segments"], st["sphere_tests"] / st["segments"])
        rows.append({"config": name, "spp_run": spp, "kernel_ms": round(st["kernel_ms"], 3), "Msamples/s": round(st["samples"] / st["kernel_ms"] / 1e3, 1),
                     "segments_per_sample": round(sbar, 3), "flop_per_sample": round(f, 1),
                     "fp32_valu_frac": round(st["samples"] * f / (st["kernel_ms"] * 1e-3) / 1e12 / PEAK, 4)})
    return rows


def main():
    r, O = pt.Renderer(0), po.Oracle()
    rep = {"gate1_per_ray [equal, compared]": gate1(r, O), "gate2_converged_image_vs_reference_renderer": gate2(r), "gate3_slices_vs_one_piece": gate3(r),
           "throughput_one_gpu": throughput(r)}
    threads, host = bench.host_threads()
    rep["cpu_reference_C1_full"] = bench.cpu_baseline(1, 800, 450, 5, 32, threads)
    rep["cpu_reference_C1_full"]["host"] = host
    r.close()
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
