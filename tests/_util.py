"""Helpers shared by the tests: scene conversion, tapes, bit comparison."""
import numpy as np

import pathtracer_amd as pt
from oracle import pyoracle as po


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def bit_equal(a, b):
    return bool((bits(a) == bits(b)).all())


def same_rows(a, b):
    """Per-row bit equality for 2-D float32 arrays (NaN == NaN when the bits agree)."""
    return (bits(a) == bits(b)).reshape(len(a), -1).all(axis=1)


def to_pt_scene(sa: po.SceneArrays) -> pt.Scene:
    vel = sa.vel if np.any(sa.vel != 0) else None
    return pt.Scene(sa.pos_r, vel, sa.mtype, sa.rgba, sa.param, sa.pl_n_d, sa.pl_type, sa.pl_rgba, sa.pl_param, sa.t0, sa.t1, sa.camera)


def to_scene_arrays(s: pt.Scene) -> po.SceneArrays:
    return po.SceneArrays(s.pos_r, s.vel, s.mtype, s.rgba, s.param, s.pl_n_d, s.pl_type, s.pl_rgba, s.pl_param, s.t0, s.t1, s.camera)


def builtin_scene_arrays(scene_id, aspect=16 / 9, n=0) -> po.SceneArrays:
    """Scene from the PRODUCT's host-side builders (pinned to the reference by tests/test_scenes.py)."""
    return to_scene_arrays(pt.Scene.builtin(scene_id, aspect, n))


def rel_err(a, b, floor=1e-30):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def psnr(img, ref, peak=1.0):
    mse = float(np.mean((np.asarray(img, np.float64) - np.asarray(ref, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def mean_rel_error(img, ref, eps=1e-2):
    """Per-pixel mean relative error on linear RGB, the image gate of BASELINE.json (eps guards black pixels)."""
    img, ref = np.asarray(img, np.float64)[..., :3], np.asarray(ref, np.float64)[..., :3]
    return float(np.mean(np.abs(img - ref) / (np.abs(ref) + eps)))


def in_threads(fn, arrays, nthreads=None):
    """Run fn(*slices) over row-slices of `arrays` on several threads (the oracle's loops are single-threaded C and ctypes
    releases the GIL); returns the outputs concatenated row-wise.  For the brute-force oracle over 10^6 spheres."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = len(arrays[0])
    if nthreads is None:
        nthreads = max(1, min(16, len(os.sched_getaffinity(0))))
    cuts = np.linspace(0, n, nthreads + 1).astype(int)
    spans = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    with ThreadPoolExecutor(len(spans)) as ex:
        parts = list(ex.map(lambda ab: fn(*[x[ab[0]:ab[1]] for x in arrays]), spans))
    return tuple(np.concatenate([p[k] for p in parts], 0) for k in range(len(parts[0])))
