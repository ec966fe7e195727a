#!/usr/bin/env python3
"""Ad-hoc parity sweep: the default (tree in LDS) closest-hit query against the brute-force oracle on millions of rays of
the reference's scenes.  usage: tools/parity_sweep.py [scene id] [million rays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracer_amd as pt
from oracle import pyoracle as po
from tests import _gen
from tests._util import builtin_scene_arrays, to_pt_scene
sid = int(sys.argv[1]) if len(sys.argv) > 1 else 1
total = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 2000000
sa = builtin_scene_arrays(sid, 16 / 9)
r = pt.Renderer(0); r.upload(to_pt_scene(sa))
O = po.Oracle(); h = O.scene(sa)
rng = np.random.default_rng(123)
bad = done = 0
while done < total:
    n = min(250000, total - done)
    uv = rng.uniform(0, 1, size=(n, 2)).astype(np.float32)
    cam_rays, _, _ = O.get_rays(sa.camera, 0, uv)
    rays, times = _gen.scene_rays(rng, sa, n, cam_rays)
    t, prim, p, nr = r.probe_hits(rays, times, 0)
    to, po_, pto, nro = O.closest_hits(h, rays, times, po.QUERY_CLOSEST)
    same = (prim == po_) & (t.view(np.uint32) == to.view(np.uint32))
    bad += int((~same).sum()); done += n
    print(f"scene {sid}: {done} rays, {bad} differing records", flush=True)
