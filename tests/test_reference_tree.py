"""The product's reference-topology tree (pt_reference_bvh: build_bvh_tree's recipe, main.cpp:93-159) and the query
that traverses it (PT_FLAG_REFERENCE_QUERY: main.cpp:53-91 incl. the pairwise combine).

CPU part: the exporter against the trees the REAL reference built (tests/golden/paths_*.npz, generated through the
reference's own sort_render_objects / random_u32 / get_bounding_box / bounding_box), and live against oracle/_ref with
a non-zero position in the main thread's RNG stream.  GPU part: the traversal against the reference's own query
results (golden q_t / q_obj) and against the oracle's restatement of it, bit for bit."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from oracle import pyoracle as po
from tests._util import bit_equal, to_pt_scene, to_scene_arrays, psnr, mean_rel_error

GOLDEN = {"scene1": 1, "scene3": 3, "materials": 4, "random_moving": None}


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"paths_{name}.npz"))
    return z, po.SceneArrays.from_npz(z)


@pytest.mark.parametrize("name", list(GOLDEN))
def test_exporter_rebuilds_the_reference_tree_from_the_sorted_scene(golden_dir, name):
    """The fixture holds the spheres in the order the reference's build LEFT World::objects; building from that order
    must give the same tree again (the sorts find everything in place), node for node, box bit for box bit."""
    z, sa = _golden(golden_dir, name)
    boxes, links, order, depth = to_pt_scene(sa).reference_bvh(0)
    gb, gl = z["boxes"].reshape(-1, 6), z["links"].reshape(-1, 3)
    assert len(boxes) == len(gb) == 2 * sa.n_spheres - 1
    assert (order == np.arange(sa.n_spheres)).all()
    assert (links == gl).all()
    assert bit_equal(boxes, gb)
    assert 2 ** depth >= sa.n_spheres


@pytest.mark.parametrize("name,sid", [(k, v) for k, v in GOLDEN.items() if v])
def test_exporter_from_the_unsorted_builtin_scene(golden_dir, name, sid):
    """From the scene as the builders lay it out (the order main() hands to build_bvh_tree): same boxes, same links, and
    `order` is the permutation the reference's in-place quicksort -- NOT stable; book-1 has 256 spheres with the same
    y -- applies to World::objects."""
    z, sa = _golden(golden_dir, name)
    scene = pt.Scene.builtin(sid, 16 / 9)
    boxes, links, order, _ = scene.reference_bvh(0)
    gb, gl = z["boxes"].reshape(-1, 6), z["links"].reshape(-1, 3)
    assert bit_equal(boxes, gb) and (links[:, :2] == gl[:, :2]).all()
    leaf = gl[:, 2] >= 0
    assert (links[leaf, 2] == order[gl[leaf, 2]]).all() and (links[~leaf, 2] == -1).all()
    assert bit_equal(scene.pos_r[order], sa.pos_r)


def test_exporter_limits():
    s = pt.Scene.builtin(5, 16 / 9, pt.REFERENCE_MAX_SPHERES + 1)
    with pytest.raises(pt.PtError):
        s.reference_bvh(0)
    b, l, o, d = pt.Scene.builtin(5, 16 / 9, 1).reference_bvh(7)
    assert len(b) == 1 and (l == [[-1, -1, 0]]).all() and d == 0


@pytest.mark.ref
@pytest.mark.parametrize("sid,skip", [(1, 0), (1, 1234), (2, 77), (3, 5)])
def test_exporter_live_against_the_reference(ref, sid, skip):
    """build_bvh_tree continues the main thread's MT19937 stream wherever scene construction left it: any offset."""
    h = ref.scene_builtin(sid, 16 / 9)
    before = ref.scene_arrays(h)
    nn = ref.build_bvh(h, skip)
    gb, gl = ref.dump_bvh(h, nn)
    after = ref.scene_arrays(h)
    ref.scene_free(h)
    boxes, links, order, _ = to_pt_scene(before).reference_bvh(skip)
    assert bit_equal(boxes, gb) and (links[:, :2] == gl[:, :2]).all()
    leaf = gl[:, 2] >= 0
    assert (links[leaf, 2] == order[gl[leaf, 2]]).all()
    assert bit_equal(before.pos_r[order], after.pos_r)


# ---------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GOLDEN))
def test_reference_query_bit_equal_to_the_reference(renderer, oracle, golden_dir, name):
    z, sa = _golden(golden_dir, name)
    renderer.upload(to_pt_scene(sa))
    renderer.build_reference_tree(0)
    rays, times = z["rays"], z["times"]
    t, prim, p, nr = renderer.probe_hits(rays, times, pt.FLAG_REFERENCE_QUERY)
    # (1) the reference's own answers: wherever the frame's closest hit is a sphere, it is the BVH query's (t, object)
    qt, qobj = z["q_t"], z["q_obj"]
    on_sphere = prim >= 0
    assert on_sphere.sum() > len(rays) // 16
    assert bit_equal(t[on_sphere], qt[on_sphere]) and (prim[on_sphere] == qobj[on_sphere]).all()
    # ... and wherever the reference's query reports a valid sphere hit in front of every plane, the frame takes it
    h = oracle.scene(sa)
    oracle.scene_set_bvh(h, z["boxes"], z["links"])
    try:
        to, po_, pto, nro = oracle.closest_hits(h, rays, times, po.QUERY_BVH_QUIRK)
        assert (prim == po_).all(), int((prim != po_).sum())
        assert bit_equal(t, to) and bit_equal(p, pto) and bit_equal(nr, nro)
        # the mode differs from the default exactly where the quirk bites (secondary rays that start on a sphere)
        td, primd, _, _ = renderer.probe_hits(rays, times, 0)
        print(f"{name}: {int((primd != prim).sum())} of {len(rays)} rays answered differently by the closest-hit default")
        # (2) whole samples and a frame, against the oracle in quirk mode
        rng = np.random.default_rng(3)
        w, hgt, n = 160, 90, 1500
        pixels, samples = rng.integers(0, w * hgt, n).astype(np.uint32), rng.integers(0, 999, n).astype(np.uint32)
        c, segs = renderer.probe_samples(w, hgt, pixels, samples, 50, seed=9, flags=pt.FLAG_REFERENCE_QUERY)
        co, segso = oracle.samples_philox(h, w, hgt, pixels, samples, 50, seed=9, query=po.QUERY_BVH_QUIRK)
        assert (segs == segso).all() and bit_equal(c, co)
        spp = 10
        img = renderer.render(56, 32, 0, spp, 50, seed=4, flags=pt.FLAG_REFERENCE_QUERY)
        st = renderer.stats()
        ref_img, seg = oracle.render_philox(h, 56, 32, 0, spp, 50, seed=4, query=po.QUERY_BVH_QUIRK)
        assert st["kernel"] == 4 and st["segments"] == seg
        assert np.abs(img.astype(np.float64) - ref_img.astype(np.float64)).max() <= 2 * spp * 2.0 ** -24 * max(1.0, float(np.abs(ref_img).max()))
    finally:
        oracle.scene_free(h)


@pytest.mark.gpu
def test_reference_query_flag_combinations(renderer):
    renderer.upload(pt.Scene.builtin(3, 1.0))
    for bad in (pt.FLAG_FORCE_BVH, pt.FLAG_FORCE_LDS, pt.FLAG_BRUTE_FORCE, pt.FLAG_STRICT_SEARCH):
        with pytest.raises(pt.PtError) as e:
            renderer.render(8, 8, 0, 1, 5, flags=pt.FLAG_REFERENCE_QUERY | bad)
        assert e.value.status == pt.PT_ERR_UNSUPPORTED
    img = renderer.render(16, 16, 0, 2, 5, flags=pt.FLAG_REFERENCE_QUERY)      # tree built on first use (rng_skip 0)
    assert np.allclose(img[..., 3], 2, rtol=1e-6) and renderer.stats()["kernel"] == 4
    renderer.upload(pt.Scene.builtin(5, 16 / 9, pt.REFERENCE_MAX_SPHERES + 1))
    with pytest.raises(pt.PtError) as e:
        renderer.render(8, 8, 0, 1, 5, flags=pt.FLAG_REFERENCE_QUERY)
    assert e.value.status == pt.PT_ERR_UNSUPPORTED


@pytest.mark.gpu
def test_converged_image_in_reference_query_mode_reaches_the_noise_floor(renderer, golden_dir):
    """With the reference's tree and its combine the converged image agrees with the reference renderer as well as a
    second reference run does (the floor stored with the fixture); the closest-hit default sits 6 dB below it (D1)."""
    z = np.load(os.path.join(golden_dir, "converged_images.npz"))
    for key, sid in (("scene1", 1), ("scene3", 3)):
        ref_img, w, h, spp, depth = z[key + "_img"], int(z[key + "_w"]), int(z[key + "_h"]), int(z[key + "_spp"]), int(z[key + "_depth"])
        floor_psnr, floor_mre = float(z[key + "_floor_psnr"]), float(z[key + "_floor_mre"])
        renderer.upload(pt.Scene.builtin(sid, w / h))
        renderer.build_reference_tree(0)                                    # the fixture's trees were built on a fresh stream
        img = pt.resolve(renderer.render(w, h, 0, spp, depth, seed=321, flags=pt.FLAG_REFERENCE_QUERY), spp).reshape(h, w, 4)
        p, m = psnr(img[..., :3], ref_img[..., :3]), mean_rel_error(img, ref_img)
        print(f"{key}: reference-query mode PSNR {p:.2f} dB (floor {floor_psnr:.2f}), mean rel err {m:.5f} (floor {floor_mre:.5f})")
        assert p >= floor_psnr - 0.75, (key, p, floor_psnr)
        assert m <= floor_mre * 1.1, (key, m, floor_mre)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GOLDEN))
def test_whole_paths_in_reference_query_mode_against_the_reference(renderer, oracle, golden_dir, name):
    """cast_ray (main.cpp:171-307) end to end: with the reference's tree, its combine and each ray's slice of the
    reference's own MT19937 streams as tape, EVERY path's radiance is the reference's -- the only difference left is the
    multiplication order of the colours (D2: loop vs recursion), <= 1e-5 relative; against the oracle in the same mode
    and order it is bit-equal.  (In the default closest-hit mode this holds only for the paths D1 does not touch.)"""
    z, sa = _golden(golden_dir, name)
    renderer.upload(to_pt_scene(sa))
    renderer.build_reference_tree(0)
    h = oracle.scene(sa)
    oracle.scene_set_bvh(h, z["boxes"], z["links"])
    try:
        rays, times = z["rays"], z["times"]
        for depth in (1, 5, 50):
            ref = z[f"rgba_d{depth}"]
            c_ref, _, _, o32, o64, _ = oracle.cast_rays_mt(h, rays, times, depth, 11, 3, po.QUERY_BVH_QUIRK, po.PRODUCT_RECURSIVE)
            assert bit_equal(c_ref, ref)
            t32, t64 = oracle.mt_f32(11, int(o32[-1]) + 64), oracle.mt_f64(3, int(o64[-1]) + 8)
            c_gpu, segs = renderer.probe_paths(rays, times, depth, t32, o32, t64, o64, pt.FLAG_REFERENCE_QUERY)
            assert ((segs >> 31) == 0).all()                                   # same decisions -> same draws: no tape overrun
            c_it, over, _, _ = oracle.cast_rays_tape(h, rays, times, depth, t32, o32, t64, o64, po.QUERY_BVH_QUIRK, po.PRODUCT_ITERATIVE)
            assert (over == 0).all() and bit_equal(c_gpu, c_it), (name, depth)
            rel = np.abs(c_gpu.astype(np.float64) - ref.astype(np.float64)) / np.maximum(np.abs(ref.astype(np.float64)), 1e-6)
            assert rel.max() <= 1e-5, (name, depth, float(rel.max()))
    finally:
        oracle.scene_free(h)
