"""Deterministic input generators shared by the tests and tests/golden/make_golden.py."""
import numpy as np

F32_MAX = np.float32(3.402823466e+38)


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=1, keepdims=True))


def random_rays(rng, n, spread=5.0):
    o = rng.normal(size=(n, 3)) * spread
    d = unit(rng.normal(size=(n, 3)))
    rays = np.concatenate([o, d], 1).astype(np.float32)
    # re-normalise in f32 the way the reference's Ray ctor would leave a normalised dir: good enough as "unit" input
    return rays


def sphere_cases(rng, n):
    """(ray, sphere) pairs: generic, aimed-at (hits), tangent-ish, origin inside, sphere behind."""
    rays = random_rays(rng, n)
    c = rng.normal(size=(n, 3)) * 5
    r = rng.uniform(0.1, 4.0, size=(n, 1))
    k = n // 5
    # aimed: direction towards a point inside the sphere
    tgt = c[:k] + unit(rng.normal(size=(k, 3))) * r[:k] * rng.uniform(0, 0.999, size=(k, 1))
    rays[:k, 3:] = unit(tgt - rays[:k, :3])
    # tangent-ish: aim at a point at distance r*(1 +- 1e-4) from the centre, perpendicular to the view direction
    v = unit(c[k:2 * k] - rays[k:2 * k, :3])
    perp = unit(np.cross(v, rng.normal(size=(k, 3))))
    tgt = c[k:2 * k] + perp * r[k:2 * k] * (1 + rng.uniform(-1e-4, 1e-4, size=(k, 1)))
    rays[k:2 * k, 3:] = unit(tgt - rays[k:2 * k, :3])
    # origin inside the sphere
    rays[2 * k:3 * k, :3] = c[2 * k:3 * k] + unit(rng.normal(size=(k, 3))) * r[2 * k:3 * k] * rng.uniform(0, 0.99, size=(k, 1))
    # sphere behind the ray
    rays[3 * k:4 * k, 3:] = -unit(c[3 * k:4 * k] - rays[3 * k:4 * k, :3])
    rays[:, 3:] = unit(rays[:, 3:])
    return rays.astype(np.float32), np.concatenate([c, r], 1).astype(np.float32)


def plane_cases(rng, n):
    rays = random_rays(rng, n)
    nrm = unit(rng.normal(size=(n, 3)))
    d = rng.normal(size=(n, 1)) * 3
    k = n // 8
    nrm[:k] = np.array([0.0, 1.0, 0.0])
    rays[:k, 4] = 0.0  # exactly parallel to the y=const plane
    rays[:k, 3:] = unit(rays[:k, 3:] + np.array([1e-30, 0, 0]))
    return rays.astype(np.float32), np.concatenate([nrm, d], 1).astype(np.float32)


def box_cases(rng, n):
    rays = random_rays(rng, n)
    boxes = np.concatenate([rng.normal(size=(n, 3)) * 5, rng.uniform(0.1, 6, size=(n, 3))], 1)
    k = n // 4
    axis = rng.integers(0, 3, k)
    rays[:k, 3:] = 0.0
    rays[np.arange(k), 3 + axis] = rng.choice([-1.0, 1.0], k)  # axis-aligned: 1/0 -> inf, 0*inf -> NaN paths
    return rays.astype(np.float32), boxes.astype(np.float32)


def scene_rays(rng, sa, n, camera_rays):
    """Rays through a scene: half camera rays, half 'secondary-like' rays leaving sphere surfaces."""
    k = n // 2
    idx = rng.integers(0, sa.n_spheres, n - k)
    nrm = unit(rng.normal(size=(n - k, 3)))
    org = sa.pos_r[idx, :3] + sa.pos_r[idx, 3:4] * nrm * (1 + 1e-4)
    d = unit(nrm + unit(rng.normal(size=(n - k, 3))) * 0.9)
    rays = np.concatenate([camera_rays[:k], np.concatenate([org, d], 1)], 0).astype(np.float32)
    t1 = sa.t1 if sa.t1 > sa.t0 else sa.t0
    times = rng.uniform(sa.t0, t1, size=n).astype(np.float32) if t1 > sa.t0 else np.full(n, sa.t0, np.float32)
    return rays, times


def random_scene(rng, n_spheres, n_planes=1, moving=False, extent=8.0):
    """A small random scene with all three materials, for fuzzing parity beyond the built-in scenes."""
    from oracle.pyoracle import SceneArrays
    pos = rng.uniform(-extent, extent, size=(n_spheres, 3))
    pos[:, 1] = np.abs(pos[:, 1]) * 0.3 + 0.2
    rad = rng.uniform(0.2, 1.2, size=(n_spheres, 1))
    vel = rng.normal(size=(n_spheres, 3)) * 0.3 * (rng.uniform(size=(n_spheres, 1)) < 0.5) if moving else np.zeros((n_spheres, 3))
    mtype = rng.choice([1, 2, 3], n_spheres, p=[0.5, 0.3, 0.2]).astype(np.uint32)
    rgba = np.concatenate([rng.uniform(0.1, 1, size=(n_spheres, 3)), np.ones((n_spheres, 1))], 1)
    rgba[mtype == 3, :3] = 1.0
    param = np.zeros((n_spheres, 2))
    param[:, 0] = np.where(mtype == 2, rng.uniform(0, 0.6, n_spheres) * (rng.uniform(size=n_spheres) < 0.7), 0)
    param[:, 1] = np.where(mtype == 3, rng.uniform(1.2, 1.8, n_spheres), 0)
    pl_n_d = np.array([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, -extent * 1.5], [-1.0, 0.0, 0.0, -extent * 1.5]])[:n_planes]
    pl_type = np.array([1, 2, 3], np.uint32)[:n_planes]
    pl_rgba = np.array([[0.6, 0.6, 0.6, 1], [0.8, 0.7, 0.9, 1], [1, 1, 1, 1]], np.float64)[:n_planes]
    pl_param = np.array([[0, 0], [0.1, 0], [0, 1.4]], np.float64)[:n_planes]
    cam_pos = np.array([0.3, 3.0, extent * 2.2])
    cdir = unit((np.array([0.0, 0.5, 0.0]) - cam_pos)[None])[0]
    up = unit(np.array([[0.05, 1.0, 0.1]]))[0]
    camera = np.concatenate([cam_pos, cdir, up, [0.7, 16 / 9, extent * 2.0, 0.05]]).astype(np.float32)
    t0, t1 = (0.0, 1.0) if moving else (0.0, 0.0)
    return SceneArrays(np.concatenate([pos, rad], 1), vel, mtype, rgba, param, pl_n_d, pl_type, pl_rgba, pl_param, t0, t1, camera)
