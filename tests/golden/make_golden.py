#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libptref.so, built from /root/reference/src).

Run in the development container only (the reference does not exist on the GPU box):
    make -C oracle all && python tests/golden/make_golden.py
Every file holds inputs and the reference's outputs -- data, no reference code.  The reference ships no tests or
vectors of its own; these are the vectors the oracle and the HIP path are pinned against
wherever libptref.so is absent.  Sizes are kept small (a few hundred KB each).
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests import _gen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REFERENCE_OUTPUT = "/root/reference/output"


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KB")


def main():
    R, O = po.Ref(), po.Oracle()
    rng = np.random.default_rng(20261111)

    # ---- 1. primitives (geometry.cpp:39-115, geometry.h:62-67, vectors.cpp:127-130, main.cpp:161-168)
    sr, ss = _gen.sphere_cases(rng, 4096)
    pr, pp = _gen.plane_cases(rng, 1024)
    br, bb = _gen.box_cases(rng, 1024)
    dirs = _gen.unit(rng.normal(size=(1024, 3))).astype(np.float32)
    nrm = _gen.unit(rng.normal(size=(1024, 3))).astype(np.float32)
    vecs = (rng.normal(size=(1024, 3)) * 10.0 ** rng.uniform(-3, 3, size=(1024, 1))).astype(np.float32)
    cr = rng.uniform(-1, 1, size=(256, 2))
    cr[:, 1] = rng.uniform(0.3, 3.0, size=256)
    save("prims.npz", sphere_rays=sr, spheres=ss, sphere_t=R.intersect_spheres(sr, ss),
         plane_rays=pr, planes=pp, plane_t=R.intersect_planes(pr, pp),
         box_rays=br, boxes=bb, box_hit=R.hit_boxes(br, bb),
         dirs=dirs, normals=nrm, reflected=R.reflect(dirs, nrm), vecs=vecs, normalized=R.normalize(vecs),
         refl_in=cr, refl_out=np.array([R.reflectance(c, r) for c, r in cr]), sizeofs=np.array(list(R.sizeofs().values()), np.uint32))

    # ---- 2. RNG stream and samplers (utils.h:27-138)
    us, us_next = R.unit_sphere_points(0, 1024)
    uv_, uv_next = R.unit_vectors(3, 1024)
    uc, uc_next = R.unit_circle_points(5, 1024)
    save("rng.npz", f32=R.random_f32(0, 4096), f64=R.random_f64(0, 1024), unit_sphere=us, unit_sphere_next=us_next,
         unit_vectors=uv_, unit_vectors_next=uv_next, unit_circle=uc, unit_circle_next=uc_next)

    # ---- 3. scenes (scene_init.cpp:35-177) as the reference builds them under libstdc++, + BMP headers of its outputs
    d = {}
    for sid in (1, 2, 3):
        h = R.scene_builtin(sid, 16 / 9)
        d.update(R.scene_arrays(h).to_npz_dict(f"s{sid}_"))
        R.scene_free(h)
    heads, dims, sizes, names = [], [], [], []
    for f in sorted(glob.glob(os.path.join(REFERENCE_OUTPUT, "*.bmp"))):
        b = np.fromfile(f, np.uint8)
        heads.append(b[:58].copy())
        dims.append((int(b[18:22].view(np.uint32)[0]), int(b[22:26].view(np.int32)[0])))
        sizes.append(len(b))
        names.append(os.path.basename(f))
    save("scenes.npz", bmp_headers=np.array(heads), bmp_dims=np.array(dims, np.int64), bmp_sizes=np.array(sizes, np.int64),
         bmp_names=np.array(names), **d)

    # ---- 4. Camera::get_ray (camera.cpp:57-79): 1024 consecutive calls after 3 skipped draws, per scene camera
    d = {}
    for sid in (1, 2, 3):
        cam = d[f"cam{sid}"] = np.load(os.path.join(OUT, "scenes.npz"))[f"s{sid}_camera"]
        uv = d[f"uv{sid}"] = rng.uniform(-0.05, 1.05, size=(1024, 2)).astype(np.float32)
        d[f"rays{sid}"], d[f"next{sid}"] = R.get_rays(cam, 3, uv)
    save("cameras.npz", **d)

    # ---- 5. BVH query + cast_ray (main.cpp:53-91, 171-307) on reference-topology trees
    scenes = {"scene1": None, "scene3": None,
              "random_moving": _gen.random_scene(np.random.default_rng(7), 24, n_planes=3, moving=True),
              "materials": None}
    for name in scenes:
        if name == "scene1":
            h = R.scene_builtin(1, 16 / 9)
        elif name == "scene3":
            h = R.scene_builtin(3, 16 / 9)
        elif name == "materials":
            import pathtracer_amd as pt
            from tests._util import to_scene_arrays
            h = R.scene_from_arrays(to_scene_arrays(pt.Scene.builtin(4, 16 / 9)))
        else:
            h = R.scene_from_arrays(scenes[name])
        nn = R.build_bvh(h, 0)
        boxes, links = R.dump_bvh(h, nn)
        sa = R.scene_arrays(h)  # in the order the BVH build left world.objects
        uv = rng.uniform(0, 1, size=(1024, 2)).astype(np.float32)
        cam_rays, _ = R.get_rays(sa.camera, 0, uv)
        rays, times = _gen.scene_rays(rng, sa, 1024, cam_rays)
        qt, qobj = R.bvh_query(h, rays, times)
        dd = dict(sa.to_npz_dict(), boxes=boxes, links=links, rays=rays, times=times, q_t=qt, q_obj=qobj)
        for depth in (1, 5, 50):
            c, n32, n64 = R.cast_rays(h, rays, times, depth, 11, 3)
            dd[f"rgba_d{depth}"], dd[f"next32_d{depth}"], dd[f"next64_d{depth}"] = c, n32, np.float64(n64)
        # the spp loop of run_thread_batch (main.cpp:341-361) on a small rect
        dd["rect_img"] = R.render_rect(h, 40, 24, 3, 5, (4, 2, 36, 20), 5, 1)
        save(f"paths_{name}.npz", **dd)
        R.scene_free(h)

    # ---- 6. converged reference renders for the image gate + the reference-vs-reference noise floor
    from tests._util import psnr, mean_rel_error
    d = {}
    for key, sid, w, hgt, spp, depth in (("scene1", 1, 160, 90, 4096, 50), ("scene3", 3, 160, 90, 4096, 50)):
        h = R.scene_builtin(sid, w / hgt)
        R.build_bvh(h, 0)
        _, a = R.render(h, w, hgt, spp, depth, 8, 0)
        _, b = R.render(h, w, hgt, spp, depth, 8, 1000003)
        R.scene_free(h)
        d[key + "_img"], d[key + "_w"], d[key + "_h"], d[key + "_spp"], d[key + "_depth"] = a, w, hgt, spp, depth
        d[key + "_floor_psnr"], d[key + "_floor_mre"] = psnr(b[..., :3], a[..., :3]), mean_rel_error(b, a)
        print(key, "noise floor: PSNR", d[key + "_floor_psnr"], "mean rel err", d[key + "_floor_mre"])
    save("converged_images.npz", **d)


if __name__ == "__main__":
    main()
