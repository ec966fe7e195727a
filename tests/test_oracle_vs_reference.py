"""CPU: the oracle restatement against the compiled reference LIVE (oracle/_ref/libptref.so) on fresh random inputs --
more cases than the committed fixtures hold.  Skipped where the library is absent (then tests/test_oracle_golden.py
carries the pin)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import _gen
from tests._util import bit_equal

pytestmark = pytest.mark.ref


def test_rng_streams(oracle, ref):
    for skip in (0, 1, 623, 624, 625, 5000):  # around the MT19937 state refresh
        assert bit_equal(oracle.mt_f32(skip, 700), ref.random_f32(skip, 700))
    assert (oracle.mt_f64(311, 700) == ref.random_f64(311, 700)).all()


def test_primitives_fuzz(oracle, ref):
    rng = np.random.default_rng(99)
    for _ in range(3):
        r, s = _gen.sphere_cases(rng, 20000)
        assert bit_equal(oracle.intersect_spheres(r, s), ref.intersect_spheres(r, s))
        r, p = _gen.plane_cases(rng, 8000)
        assert bit_equal(oracle.intersect_planes(r, p), ref.intersect_planes(r, p))
        r, b = _gen.box_cases(rng, 8000)
        assert (oracle.hit_boxes(r, b) == ref.hit_boxes(r, b)).all()
    d, n = _gen.unit(rng.normal(size=(5000, 3))).astype(np.float32), rng.normal(size=(5000, 3)).astype(np.float32)
    assert bit_equal(oracle.reflect(d, n), ref.reflect(d, n))
    assert bit_equal(oracle.normalize(n), ref.normalize(n))
    assert all(oracle.reflectance(c, r) == ref.reflectance(c, r) for c, r in rng.uniform(-1.2, 1.2, size=(2000, 2)))


def test_samplers(oracle, ref):
    for fn in ("unit_sphere_points", "unit_vectors", "unit_circle_points"):
        for skip in (0, 7, 1234):
            (a, na), (b, nb) = getattr(oracle, fn)(skip, 4000), getattr(ref, fn)(skip, 4000)
            assert bit_equal(a, b) and na == nb, (fn, skip)


def test_camera_construction_and_get_ray(oracle, ref):
    rng = np.random.default_rng(5)
    for _ in range(8):
        pos, tgt = rng.normal(size=3) * 10, rng.normal(size=3) * 3
        up = _gen.unit(rng.normal(size=(1, 3)) * 0.3 + np.array([[0, 1, 0]]))[0]
        lens = (float(rng.uniform(0, 1)), float(rng.uniform(1, 40))) if rng.uniform() < 0.7 else None
        cam = ref.camera_make(pos, float(rng.uniform(20, 120)), float(rng.uniform(0.5, 2.4)), tgt, up, lens)
        uv = rng.uniform(-0.1, 1.1, size=(3000, 2)).astype(np.float32)
        (a, na, _), (b, nb) = oracle.get_rays(cam, 2, uv), ref.get_rays(cam, 2, uv)
        assert bit_equal(a, b) and na == nb


@pytest.mark.parametrize("sid,aspect", [(1, 800 / 450), (2, 16 / 9), (3, 16 / 9)])
def test_builtin_scenes_bvh_cast_ray_render(oracle, ref, sid, aspect):
    rng = np.random.default_rng(sid)
    h = ref.scene_builtin(sid, aspect)
    nn = ref.build_bvh(h, 0)
    boxes, links = ref.dump_bvh(h, nn)
    sa = ref.scene_arrays(h)
    assert nn == 2 * sa.n_spheres - 1
    oh = oracle.scene(sa)
    oracle.scene_set_bvh(oh, boxes, links)
    uv = rng.uniform(0, 1, size=(2000, 2)).astype(np.float32)
    cam_rays, _ = ref.get_rays(sa.camera, 0, uv)
    rays, times = _gen.scene_rays(rng, sa, 3000, cam_rays)
    (t1, o1), (t2, o2) = oracle.bvh_query(oh, rays, times, po.QUERY_BVH_QUIRK), ref.bvh_query(h, rays, times)
    assert bit_equal(t1, t2) and (o1 == o2).all()
    for depth in (1, 5, 50):
        c1, a32, a64, *_ = oracle.cast_rays_mt(oh, rays, times, depth, 17, 5)
        c2, b32, b64 = ref.cast_rays(h, rays, times, depth, 17, 5)
        assert bit_equal(c1, c2) and a32 == b32 and a64 == b64, depth
    a = oracle.render_rect_mt(oh, 48, 27, 3, 5, (5, 3, 40, 22), 9, 2)
    b = ref.render_rect(h, 48, 27, 3, 5, (5, 3, 40, 22), 9, 2)
    assert bit_equal(a, b)
    oracle.scene_free(oh)
    ref.scene_free(h)


def test_random_scenes_all_materials(oracle, ref):
    rng = np.random.default_rng(1234)
    for k, (ns, npl, moving) in enumerate([(1, 0, False), (2, 1, True), (33, 3, True), (150, 2, False)]):
        sa0 = _gen.random_scene(rng, ns, n_planes=npl, moving=moving)
        h = ref.scene_from_arrays(sa0)
        nn = ref.build_bvh(h, k)
        boxes, links = ref.dump_bvh(h, nn)
        sa = ref.scene_arrays(h)
        oh = oracle.scene(sa)
        oracle.scene_set_bvh(oh, boxes, links)
        uv = rng.uniform(0, 1, size=(1000, 2)).astype(np.float32)
        cam_rays, _ = ref.get_rays(sa.camera, 0, uv)
        rays, times = _gen.scene_rays(rng, sa, 2000, cam_rays)
        c1, a32, a64, *_ = oracle.cast_rays_mt(oh, rays, times, 50, 3, 1)
        c2, b32, b64 = ref.cast_rays(h, rays, times, 50, 3, 1)
        assert bit_equal(c1, c2) and a32 == b32 and a64 == b64, (ns, npl, moving)
        oracle.scene_free(oh)
        ref.scene_free(h)
