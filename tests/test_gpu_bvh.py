"""GPU: the BVH-in-HBM kernel family (scenes that overflow LDS, or PT_FLAG_FORCE_BVH) against the oracle.
Leaf tests run in reference arithmetic and the query is the true closest hit, so the bar is bit-exact, whatever the tree."""
import numpy as np
import pytest

import pathtracer_amd as pt
from oracle import pyoracle as po
from tests import _gen
from tests._util import bit_equal, to_pt_scene, builtin_scene_arrays, in_threads

pytestmark = pytest.mark.gpu


def big_scenes():
    rng = np.random.default_rng(77)
    return {
        "scene2_2400": (builtin_scene_arrays(2, 16 / 9), 0),                   # overflows LDS -> BVH chosen automatically
        "random5000": (builtin_scene_arrays(5, 16 / 9, 5000), 0),
        "scene1_forced": (builtin_scene_arrays(1, 16 / 9), pt.FLAG_FORCE_BVH),
        "scene1_default_lds_tree": (builtin_scene_arrays(1, 16 / 9), 0),
        "random_moving_300_default_lds_tree": (_gen.random_scene(np.random.default_rng(78), 300, n_planes=2, moving=True, extent=14.0), 0),
        "tiny_2_default": (_gen.random_scene(np.random.default_rng(79), 2, n_planes=1), 0),
        "scene3_moving_forced": (builtin_scene_arrays(3, 16 / 9), pt.FLAG_FORCE_BVH),
        "random_moving_300_forced": (_gen.random_scene(rng, 300, n_planes=2, moving=True, extent=14.0), pt.FLAG_FORCE_BVH),
        "tiny_3_forced": (_gen.random_scene(rng, 3, n_planes=1), pt.FLAG_FORCE_BVH),      # root is a leaf
        "planes_only_forced": (_gen.random_scene(rng, 0, n_planes=2), pt.FLAG_FORCE_BVH),
    }


SCENES = big_scenes()


@pytest.fixture(scope="module", params=list(SCENES))
def scene(request, renderer, oracle):
    sa, flags = SCENES[request.param]
    renderer.upload(to_pt_scene(sa))
    h = oracle.scene(sa)
    yield request.param, sa, h, flags
    oracle.scene_free(h)


def _rays(oracle, sa, n, seed):
    rng = np.random.default_rng(seed)
    uv = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
    cam_rays, _, _ = oracle.get_rays(sa.camera, 0, uv)
    if sa.n_spheres == 0:
        return cam_rays, np.full(n, sa.t0, np.float32)
    return _gen.scene_rays(rng, sa, n, cam_rays)


def test_bvh_hits_bit_exact(scene, renderer, oracle):
    name, sa, h, flags = scene
    rays, times = _rays(oracle, sa, 8192, 5)
    t, prim, p, nr = renderer.probe_hits(rays, times, flags)
    to, po_, pto, nro = oracle.closest_hits(h, rays, times, po.QUERY_CLOSEST)
    assert (prim == po_).all(), (name, int((prim != po_).sum()))
    assert bit_equal(t, to) and bit_equal(p, pto) and bit_equal(nr, nro), name


def test_bvh_samples_bit_exact(scene, renderer, oracle):
    name, sa, h, flags = scene
    rng = np.random.default_rng(6)
    w, hgt, n = 160, 90, 3000
    pixels, samples = rng.integers(0, w * hgt, n).astype(np.uint32), rng.integers(0, 9999, n).astype(np.uint32)
    c, segs = renderer.probe_samples(w, hgt, pixels, samples, 50, seed=8, flags=flags)
    co, segso = oracle.samples_philox(h, w, hgt, pixels, samples, 50, seed=8)
    assert (segs == segso).all(), name
    assert bit_equal(c, co), name


def test_bvh_render_equals_oracle(scene, renderer, oracle):
    name, sa, h, flags = scene
    w, hgt, spp, depth = 52, 30, 3, 12
    img = renderer.render(w, hgt, 0, spp, depth, seed=4, flags=flags)
    st = renderer.stats()
    ref, seg = oracle.render_philox(h, w, hgt, 0, spp, depth, seed=4)
    assert st["kernel"] in (1, 3, 5) and st["segments"] == seg, name
    assert np.abs(img - ref).max() <= 8 * spp * 2.0 ** -24 * max(1.0, float(np.abs(ref).max()) / spp), name
    if sa.n_spheres:
        assert st["node_visits"] > 0 or sa.n_spheres <= 4
        assert st["sphere_tests"] > 0


def test_lds_and_bvh_paths_agree(renderer, oracle):
    """Same scene through both kernel families: identical per-sample radiance (both implement the closest-hit definition)."""
    sa = builtin_scene_arrays(1, 16 / 9)
    renderer.upload(to_pt_scene(sa))
    rng = np.random.default_rng(9)
    w, hgt, n = 320, 180, 20000
    pixels, samples = rng.integers(0, w * hgt, n).astype(np.uint32), rng.integers(0, 512, n).astype(np.uint32)
    a, _ = renderer.probe_samples(w, hgt, pixels, samples, 50, seed=1, flags=pt.FLAG_STRICT_SEARCH)
    b, _ = renderer.probe_samples(w, hgt, pixels, samples, 50, seed=1, flags=pt.FLAG_FORCE_BVH)
    c, _ = renderer.probe_samples(w, hgt, pixels, samples, 50, seed=1, flags=0)
    assert bit_equal(a, b) and bit_equal(a, c)
    with pytest.raises(pt.PtError) as e:   # contradictory flags
        renderer.probe_samples(w, hgt, pixels[:4], samples[:4], 5, flags=pt.FLAG_FORCE_BVH | pt.FLAG_FORCE_LDS)
    assert e.value.status == pt.PT_ERR_UNSUPPORTED
    renderer.upload(pt.Scene.builtin(2, 16 / 9))
    for flags in (pt.FLAG_BRUTE_FORCE, pt.FLAG_STRICT_SEARCH):                      # 2400 spheres: no brute-force image in LDS
        with pytest.raises(pt.PtError) as e:
            renderer.render(16, 9, 0, 1, 5, flags=flags)
        assert e.value.status == pt.PT_ERR_UNSUPPORTED
    img = renderer.render(48, 27, 0, 2, 12, seed=3)
    assert renderer.stats()["kernel"] == 1          # the tree would fit LDS only by crowding out the waves: read from HBM
    forced = renderer.render(48, 27, 0, 2, 12, seed=3, flags=pt.FLAG_FORCE_LDS)
    assert renderer.stats()["kernel"] == 3          # ... but can be forced there (materials stay in HBM), same samples
    assert np.abs(img - forced).max() <= 8 * 2 * 2.0 ** -24
    renderer.upload(pt.Scene.builtin(5, 16 / 9, 20000))
    with pytest.raises(pt.PtError) as e:                                            # 20000 spheres: no tree in LDS either
        renderer.render(16, 9, 0, 1, 5, flags=pt.FLAG_FORCE_LDS)
    assert e.value.status == pt.PT_ERR_UNSUPPORTED


def test_large_scene_reproduces_the_reference_formulas_phantom_hits(renderer, oracle):
    """100 000 spheres over a 1000-unit slab (C5 geometry; >= 65 536 spheres also exercises the multi-threaded builder).
    Out there the reference's own sphere formula (geometry.cpp:45-66) stops agreeing with geometry: b*b - 4c cancels down
    from |o-c|^2 ~ 3e5, so its discriminant carries an error of ~45 * 2^-24 * |o-c|^2 ~ 0.1-0.7 and it reports PHANTOM hits
    on spheres the ray misses by a tenth of a unit.  A brute-force search in that arithmetic (the oracle, = the definition
    of the query) collects them.  The tree-in-HBM kernel therefore tests boxes against a CONE around the ray (half-angle
    1e-3 >= the formula's worst case 8.2e-4) and culls against 1.002 x the closest t; with plain boxes 2e-4 of these
    rays differed.  The bar: bit-equal records; should a difference ever appear it must at least be one the formula's
    cancellation explains (checked in f64 below), never a traversal error."""
    sa = builtin_scene_arrays(5, 16 / 9, 100000)
    renderer.upload(to_pt_scene(sa))
    h = oracle.scene(sa)
    rng = np.random.default_rng(6)
    n = 40000
    uv = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
    cam_rays, _, _ = oracle.get_rays(sa.camera, 0, uv)
    rays, times = _gen.scene_rays(rng, sa, n, cam_rays)
    t, prim, p, nr = renderer.probe_hits(rays, times, 0)
    to, po_, pto, nro = oracle.closest_hits(h, rays, times, po.QUERY_CLOSEST)
    oracle.scene_free(h)
    same = prim == po_
    assert bit_equal(t[same], to[same]) and bit_equal(p[same], pto[same]) and bit_equal(nr[same], nro[same])
    bad = np.nonzero(~same)[0]
    print(f"large scene: {len(bad)} of {n} records differ from the brute-force oracle")
    assert len(bad) <= 1e-4 * n, len(bad)

    def conditioning(i, sphere):           # (f64 quarter discriminant, error bound of the f32 formula) for ray i
        c = sa.pos_r[sphere].astype(np.float64)
        oc = rays[i, :3].astype(np.float64) - c[:3]
        d = rays[i, 3:].astype(np.float64)
        hb = oc @ d
        return hb * hb - (oc @ oc - c[3] ** 2), 45 * 2.0 ** -24 * (oc @ oc)

    for i in bad:
        # the two answers differ -> at least one of the spheres involved must sit inside the formula's error band:
        # a phantom (or barely real) hit, or two hits whose t's are closer than the formula resolves
        bands = [conditioning(i, s) for s in (prim[i], po_[i]) if s >= 0]
        phantom = any(disc <= err for disc, err in bands)
        close = prim[i] >= 0 and po_[i] >= 0 and abs(float(t[i]) - float(to[i])) <= 2e-3 * float(to[i])
        assert phantom or close, (int(i), int(prim[i]), float(t[i]), int(po_[i]), float(to[i]), bands)


@pytest.mark.parametrize("kind,stack", [(1, 0), (2, 0), (3, 0), (3, 3)])
@pytest.mark.parametrize("name", ["scene2_2400", "scene1_forced", "scene3_moving_forced", "random_moving_300_forced", "tiny_3_forced"])
def test_grid_nodes_bit_exact(renderer, oracle, monkeypatch, name, kind, stack):
    """The grid-node formats of trees in HBM forced onto small scenes -- kind 1: 32-byte 2-wide nodes; 2: 64-byte 4-wide
    nodes over the binary tree's leaves; 3: 4-wide over single spheres, what big trees (>= 32768 spheres: the 10^5- and
    10^6-sphere tests) get by default; stack 3: only three levels of the 4-wide traversal's stack in LDS, so the
    per-lane scratch levels are exercised.  The boxes only decide which leaves are visited, so hits, per-sample radiance
    and rendered frames stay the oracle's, to the bit."""
    monkeypatch.setenv("PT_GRID_NODES", str(kind))
    if stack:
        monkeypatch.setenv("PT_WIDE_STACK", str(stack))
    sa, flags = SCENES[name]
    flags |= pt.FLAG_FORCE_BVH
    renderer.upload(to_pt_scene(sa))
    h = oracle.scene(sa)
    try:
        rays, times = _rays(oracle, sa, 8192, 15)
        t, prim, p, nr = renderer.probe_hits(rays, times, flags)
        to, po_, pto, nro = oracle.closest_hits(h, rays, times, po.QUERY_CLOSEST)
        assert (prim == po_).all(), (name, int((prim != po_).sum()))
        assert bit_equal(t, to) and bit_equal(p, pto) and bit_equal(nr, nro), name
        rng = np.random.default_rng(16)
        w, hgt, n = 160, 90, 2000
        pixels, samples = rng.integers(0, w * hgt, n).astype(np.uint32), rng.integers(0, 9999, n).astype(np.uint32)
        c, segs = renderer.probe_samples(w, hgt, pixels, samples, 50, seed=18, flags=flags)
        co, segso = oracle.samples_philox(h, w, hgt, pixels, samples, 50, seed=18)
        assert (segs == segso).all() and bit_equal(c, co), name
        img = renderer.render(96, 54, 0, 12, 50, seed=19, flags=flags)
        st = renderer.stats()
        ref, seg = oracle.render_philox(h, 96, 54, 0, 12, 50, seed=19)
        assert st["kernel"] == 1 and st["segments"] == seg, name
        if sa.n_spheres > 4:
            assert (st["node_bytes"], st["node_width"]) == {1: (32, 2), 2: (64, 4), 3: (64, 4)}[kind], (name, kind, st)
        assert np.abs(img.astype(np.float64) - ref.astype(np.float64)).max() <= 2 * 12 * 2.0 ** -24 * max(1.0, float(np.abs(ref).max()))
    finally:
        oracle.scene_free(h)


def test_c5_million_spheres_4k(renderer, oracle):
    """BASELINE.json config 5 at its full size: 10^6 random spheres + ground plane, 3840x2160, depth 50, through the
    default path (flags = 0: tree in HBM, 4-wide grid nodes over single spheres).  A few spp of the whole frame for the size-independent
    properties; bit-equality with the brute-force oracle (10^6 exact sphere tests per segment) on a strided subset of
    rays and of whole samples."""
    w, hgt, depth, spp = 3840, 2160, 50, 3
    sa = builtin_scene_arrays(5, w / hgt, 1000000)
    assert sa.n_spheres == 1000000
    renderer.upload(to_pt_scene(sa))
    img = renderer.render(w, hgt, 0, spp, depth, seed=12)
    st = renderer.stats()
    assert st["kernel"] == 1 and (st["node_bytes"], st["node_width"]) == (64, 4)     # tree in HBM, 4-wide grid nodes
    assert st["samples"] == w * hgt * spp
    assert np.isfinite(img).all() and (img >= 0).all()
    assert np.allclose(img[..., 3], spp, rtol=1e-6)                                  # every sample landed exactly once
    assert (img[..., :3] <= spp * 1.0000001).all()
    parts = renderer.render(w, hgt, 0, 1, depth, seed=12) + renderer.render(w, hgt, 1, spp, depth, seed=12)
    assert np.abs(img - parts).max() <= 2 * spp * 2.0 ** -24 * spp                   # linear in the sample range
    h = oracle.scene(sa)
    try:
        # (1) closest-hit records: camera rays + rays leaving sphere surfaces, all over the 1000-unit slab
        rng = np.random.default_rng(31)
        n = 3072
        uv = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
        cam_rays, _, _ = oracle.get_rays(sa.camera, 0, uv)
        rays, times = _gen.scene_rays(rng, sa, n, cam_rays)
        t, prim, p, nr = renderer.probe_hits(rays, times, 0)
        to, po_, pto, nro = in_threads(lambda r, tm: oracle.closest_hits(h, r, tm, po.QUERY_CLOSEST), (rays, times))
        assert (prim == po_).all(), int((prim != po_).sum())
        assert bit_equal(t, to) and bit_equal(p, pto) and bit_equal(nr, nro)
        assert (prim >= 0).sum() > n // 4                                            # the subset does hit spheres
        # (2) whole samples of the frame above (same seed): a strided subset of pixels, first and last sample
        ys, xs = np.meshgrid(np.arange(11, hgt, 241), np.arange(7, w, 307), indexing="ij")
        pix = np.tile((ys * w + xs).reshape(-1).astype(np.uint32), 2)
        smp = np.repeat(np.array([0, spp - 1], np.uint32), len(pix) // 2)
        c, segs = renderer.probe_samples(w, hgt, pix, smp, depth, seed=12)
        co, segso = in_threads(lambda px, sm: oracle.samples_philox(h, w, hgt, px, sm, depth, seed=12), (pix, smp))
        assert (segs == segso).all() and bit_equal(c, co)
        assert segs.mean() > 1.5
    finally:
        oracle.scene_free(h)


def _far_flung_scene(n=200, extent=1000.0, seed=41):
    """A handful of spheres spread over a thousand units: few enough for the tree-in-LDS kernel, far enough apart for the
    reference's sphere formula (geometry.cpp:45-66) to report phantom hits."""
    sa = _gen.random_scene(np.random.default_rng(seed), n, n_planes=1, extent=extent)
    sa.pos_r[:, 3] = np.random.default_rng(seed + 1).uniform(0.3, 6.0, n).astype(np.float32)
    sa.pos_r[:, 1] = sa.pos_r[:, 3] * np.random.default_rng(seed + 2).uniform(0.5, 3.0, n).astype(np.float32)
    return sa


def test_small_count_large_extent_through_the_default_kernel(renderer, oracle):
    """~200 spheres over a 2000-unit field through flags = 0: the tree fits LDS (kernel 3), and the rays work a thousand
    units from their origins, where the reference formula's cancellation (b*b - 4c from |o-c|^2 ~ 1e6) yields phantom
    hits and orders real ones by noise.  The tree-in-LDS traversal tests boxes against the same cone as the tree in HBM,
    so the answer is the brute-force oracle's, to the bit."""
    sa = _far_flung_scene()
    renderer.upload(to_pt_scene(sa))
    h = oracle.scene(sa)
    try:
        rng = np.random.default_rng(43)
        n = 60000
        # rays between the spheres' neighbourhoods: from near one sphere towards (almost) another one, so that many of
        # them graze their target inside the formula's error band at |o-c| of hundreds to a thousand units
        a, b = rng.integers(0, sa.n_spheres, n), rng.integers(0, sa.n_spheres, n)
        org = sa.pos_r[a, :3] + _gen.unit(rng.normal(size=(n, 3))) * (sa.pos_r[a, 3:4] * rng.uniform(1.001, 3.0, size=(n, 1)))
        org[:, 1] = np.abs(org[:, 1]) + 1e-3
        tgt = sa.pos_r[b, :3] + _gen.unit(rng.normal(size=(n, 3))) * (sa.pos_r[b, 3:4] * rng.uniform(0.0, 1.15, size=(n, 1)))
        d = tgt - org
        ok = np.linalg.norm(d, axis=1) > 1e-3
        rays = np.concatenate([org[ok], _gen.unit(d[ok])], 1).astype(np.float32)
        times = np.zeros(len(rays), np.float32)
        t, prim, p, nr = renderer.probe_hits(rays, times, 0)
        to, po_, pto, nro = oracle.closest_hits(h, rays, times, po.QUERY_CLOSEST)
        assert (prim == po_).all(), int((prim != po_).sum())
        assert bit_equal(t, to) and bit_equal(p, pto) and bit_equal(nr, nro)
        # the case the cone exists for is present in the sample: hits the oracle reports although the ray misses the
        # sphere geometrically (f64), i.e. phantom hits of the formula
        hit = np.nonzero(po_ >= 0)[0]
        c = sa.pos_r[po_[hit]].astype(np.float64)
        oc = rays[hit, :3].astype(np.float64) - c[:, :3]
        dd = rays[hit, 3:].astype(np.float64)
        hb = np.einsum("ij,ij->i", oc, dd)
        phantom = (hb * hb - (np.einsum("ij,ij->i", oc, oc) - c[:, 3] ** 2)) < 0
        print(f"far-flung scene: {len(hit)} sphere hits of {len(rays)} rays, {int(phantom.sum())} of them phantom hits of the formula")
        assert phantom.sum() >= 5
        # whole frames and samples through the render kernel (k_render_w7, tree in LDS)
        w, hgt, spp, depth = 96, 54, 6, 50
        img = renderer.render(w, hgt, 0, spp, depth, seed=44)
        st = renderer.stats()
        ref, seg = oracle.render_philox(h, w, hgt, 0, spp, depth, seed=44)
        assert st["kernel"] in (3, 5) and st["segments"] == seg
        assert np.abs(img.astype(np.float64) - ref.astype(np.float64)).max() <= 2 * spp * 2.0 ** -24 * max(1.0, float(np.abs(ref).max()))
    finally:
        oracle.scene_free(h)


def test_rays_returning_from_far_out_on_the_ground_plane(renderer, oracle):
    """Book-1 scene: the ground is an infinite plane, so diffuse bounces off it hundreds to ten thousand units out send
    rays back into the sphere field with |o| far beyond the scene extent -- where the slab test's rounding (|o| 2^-24)
    and the formula's phantom hits (8e-4 |o-c|) exceed any fixed box padding.  Default kernel (tree in LDS, cone test)
    against the brute-force oracle, bit for bit."""
    sa = builtin_scene_arrays(1, 16 / 9)
    renderer.upload(to_pt_scene(sa))
    h = oracle.scene(sa)
    try:
        rng = np.random.default_rng(51)
        n = 60000
        dist = 10.0 ** rng.uniform(2.0, 4.0, n)
        ang = rng.uniform(0, 2 * np.pi, n)
        org = np.stack([dist * np.cos(ang), np.full(n, 1e-3) * (1 + dist * 1e-3), dist * np.sin(ang)], 1)
        k = rng.integers(0, sa.n_spheres, n)
        tgt = sa.pos_r[k, :3] + _gen.unit(rng.normal(size=(n, 3))) * (sa.pos_r[k, 3:4] * rng.uniform(0.0, 1.5, size=(n, 1)))
        rays = np.concatenate([org, _gen.unit(tgt - org)], 1).astype(np.float32)
        times = np.zeros(n, np.float32)
        to, po_, pto, nro = oracle.closest_hits(h, rays, times, po.QUERY_CLOSEST)
        for flags in (0, pt.FLAG_FORCE_BVH):
            t, prim, p, nr = renderer.probe_hits(rays, times, flags)
            assert (prim == po_).all(), (flags, int((prim != po_).sum()))
            assert bit_equal(t, to) and bit_equal(p, pto) and bit_equal(nr, nro), flags
        assert (po_ >= 0).mean() > 0.3
    finally:
        oracle.scene_free(h)
