"""ctypes front-end for the two CHECKERS -- test infrastructure only.

  Oracle : oracle/libpt_oracle.so   our CPU restatement of the reference hot path (oracle/pt_oracle.cpp)
  Ref    : oracle/_ref/libptref.so  the REAL reference compiled from /root/reference/src (oracle/ref_harness.cpp);
                                    present only where it was built (this container) or shipped prebuilt (GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Nothing under
pathtracer_amd/ does (tests/test_product_isolation.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libpt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libptref.so")

QUERY_CLOSEST, QUERY_BVH_QUIRK = 0, 1
PRODUCT_ITERATIVE, PRODUCT_RECURSIVE = 0, 1
MAT_NONE, MAT_DIFFUSE, MAT_METAL, MAT_DIALECTRIC = 0, 1, 2, 3
EV_SCATTER, EV_SKY, EV_DEPTH, EV_ABSORB, EV_NONE_MAT = 0, 1, 2, 3, 4
PRIM_MISS = -(2 ** 31)
F32_MAX = np.float32(3.402823466e+38)


def build(force=False):
    """Compile the restatement, and the reference harness when /root/reference is present."""
    args = ["make", "-C", HERE, "all"]
    if force:
        args.insert(1, "-B")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class SceneArrays:
    """Flat scene description shared by oracle, reference harness and the HIP library's pt_scene_desc."""

    def __init__(self, pos_r, vel, mtype, rgba, param, pl_n_d, pl_type, pl_rgba, pl_param, t0, t1, camera13):
        self.pos_r = f32(pos_r).reshape(-1, 4)
        n = len(self.pos_r)
        self.vel = f32(vel).reshape(-1, 3) if vel is not None else np.zeros((n, 3), np.float32)
        self.mtype = u32(mtype).reshape(-1)
        self.rgba = f32(rgba).reshape(-1, 4)
        self.param = f32(param).reshape(-1, 2)
        self.pl_n_d = f32(pl_n_d).reshape(-1, 4)
        self.pl_type = u32(pl_type).reshape(-1)
        self.pl_rgba = f32(pl_rgba).reshape(-1, 4)
        self.pl_param = f32(pl_param).reshape(-1, 2)
        self.t0, self.t1 = float(t0), float(t1)
        self.camera = f32(camera13).reshape(13)
        assert len(self.vel) == n and len(self.mtype) == n and len(self.rgba) == n and len(self.param) == n
        assert len(self.pl_type) == len(self.pl_n_d) == len(self.pl_rgba) == len(self.pl_param)

    @property
    def n_spheres(self):
        return len(self.pos_r)

    @property
    def n_planes(self):
        return len(self.pl_n_d)

    def to_npz_dict(self, prefix=""):
        return {prefix + k: getattr(self, k) for k in
                ("pos_r", "vel", "mtype", "rgba", "param", "pl_n_d", "pl_type", "pl_rgba", "pl_param", "camera")} | {
            prefix + "t0t1": np.array([self.t0, self.t1], np.float32)}

    @staticmethod
    def from_npz(z, prefix=""):
        g = lambda k: z[prefix + k]
        t = g("t0t1")
        return SceneArrays(g("pos_r"), g("vel"), g("mtype"), g("rgba"), g("param"), g("pl_n_d"), g("pl_type"),
                           g("pl_rgba"), g("pl_param"), t[0], t[1], g("camera"))


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.oracle_scene_new.restype = C.c_void_p
        L.oracle_reflectance.restype = C.c_double
        L.oracle_reflectance.argtypes = [C.c_double, C.c_double]
        L.oracle_render_philox.restype = C.c_uint64
        L.oracle_time_render_philox.restype = C.c_double

    # ---- scenes
    def scene(self, sa: SceneArrays):
        h = self.lib.oracle_scene_new(C.c_uint32(sa.n_spheres), _p(sa.pos_r), _p(sa.vel), _p(sa.mtype), _p(sa.rgba),
                                      _p(sa.param), C.c_uint32(sa.n_planes), _p(sa.pl_n_d), _p(sa.pl_type),
                                      _p(sa.pl_rgba), _p(sa.pl_param), C.c_float(sa.t0), C.c_float(sa.t1), _p(sa.camera))
        return C.c_void_p(h)

    def scene_free(self, h):
        self.lib.oracle_scene_free(h)

    def scene_set_bvh(self, h, boxes, links):
        boxes, links = f32(boxes), np.ascontiguousarray(links, np.int32)
        self.lib.oracle_scene_set_bvh(h, C.c_uint32(len(links)), _p(boxes), _p(links))

    # ---- RNG
    def philox(self, ctr, key):
        ctr, key, out = u32(ctr), u32(key), np.zeros(4, np.uint32)
        self.lib.oracle_philox4x32_10(_p(ctr), _p(key), _p(out))
        return out

    def philox_words(self, seed, pixel, sample, n):
        out = np.zeros(n, np.uint32)
        self.lib.oracle_philox_words(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(n), _p(out))
        return out

    def philox_unit_sphere_points(self, seed, pixel, sample, n):
        out, used = np.zeros((n, 3), np.float32), C.c_uint32()
        self.lib.oracle_philox_unit_sphere_points(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(n), _p(out), C.byref(used))
        return out, int(used.value)

    def philox_f32(self, seed, pixel, sample, n):
        out = np.zeros(n, np.float32)
        self.lib.oracle_philox_f32(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(n), _p(out))
        return out

    def philox_f64(self, seed, pixel, sample, n):
        out = np.zeros(n, np.float64)
        self.lib.oracle_philox_f64(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(n), _p(out))
        return out

    def mt_f32(self, skip, n):
        out = np.zeros(n, np.float32)
        self.lib.oracle_mt_f32(C.c_uint32(skip), C.c_uint32(n), _p(out))
        return out

    def mt_f64(self, skip, n):
        out = np.zeros(n, np.float64)
        self.lib.oracle_mt_f64(C.c_uint32(skip), C.c_uint32(n), _p(out))
        return out

    # ---- primitives
    def _prim(self, fn, rays, other, out):
        rays, other = f32(rays), f32(other)
        getattr(self.lib, fn)(C.c_uint32(len(out)), _p(rays), _p(other), _p(out))
        return out

    def intersect_spheres(self, rays, spheres):
        return self._prim("oracle_intersect_spheres", rays, spheres, np.zeros(len(rays), np.float32))

    def intersect_planes(self, rays, planes):
        return self._prim("oracle_intersect_planes", rays, planes, np.zeros(len(rays), np.float32))

    def hit_boxes(self, rays, boxes):
        return self._prim("oracle_hit_boxes", rays, boxes, np.zeros(len(rays), np.uint8))

    def reflect(self, dirs, normals):
        return self._prim("oracle_reflect", dirs, normals, np.zeros((len(dirs), 3), np.float32))

    def normalize(self, v):
        v, out = f32(v), np.zeros((len(v), 3), np.float32)
        self.lib.oracle_normalize(C.c_uint32(len(v)), _p(v), _p(out))
        return out

    def reflectance(self, c, r):
        return self.lib.oracle_reflectance(float(c), float(r))

    # ---- samplers on the MT stream
    def _samp(self, fn, skip, n, dim):
        out, nxt = np.zeros((n, dim), np.float32), C.c_float()
        getattr(self.lib, fn)(C.c_uint32(skip), C.c_uint32(n), _p(out), C.byref(nxt))
        return out, np.float32(nxt.value)

    def unit_sphere_points(self, skip, n):
        return self._samp("oracle_unit_sphere_points", skip, n, 3)

    def unit_vectors(self, skip, n):
        return self._samp("oracle_unit_vectors", skip, n, 3)

    def unit_circle_points(self, skip, n):
        return self._samp("oracle_unit_circle_points", skip, n, 2)

    def get_rays(self, camera13, skip, uv):
        cam, uv = f32(camera13), f32(uv)
        n = len(uv)
        out, nxt, ofs = np.zeros((n, 6), np.float32), C.c_float(), np.zeros(n + 1, np.uint64)
        self.lib.oracle_get_rays(_p(cam), C.c_uint32(skip), C.c_uint32(n), _p(uv), _p(out), C.byref(nxt), _p(ofs))
        return out, np.float32(nxt.value), ofs

    def get_rays_tape(self, camera13, uv, tape, ofs):
        cam, uv, tape, ofs = f32(camera13), f32(uv), f32(tape), u64(ofs)
        out = np.zeros((len(uv), 6), np.float32)
        self.lib.oracle_get_rays_tape(_p(cam), C.c_uint32(len(uv)), _p(uv), _p(tape), _p(ofs), _p(out))
        return out

    # ---- queries / paths
    def bvh_query(self, h, rays, times, query):
        rays, times = f32(rays), f32(times)
        n = len(rays)
        t, obj = np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.lib.oracle_bvh_query(h, C.c_uint32(n), _p(rays), _p(times), C.c_int(query), _p(t), _p(obj))
        return t, obj

    def closest_hits(self, h, rays, times, query=QUERY_CLOSEST):
        rays, times = f32(rays), f32(times)
        n = len(rays)
        t, prim = np.zeros(n, np.float32), np.zeros(n, np.int32)
        pt, nr = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        self.lib.oracle_closest_hits(h, C.c_uint32(n), _p(rays), _p(times), C.c_int(query), _p(t), _p(prim), _p(pt), _p(nr))
        return t, prim, pt, nr

    def scatter_tape(self, dirs, points, normals, mtype, rgba, param, tape32, ofs32, tape64, ofs64):
        dirs, points, normals = f32(dirs), f32(points), f32(normals)
        mtype, rgba, param = u32(mtype), f32(rgba), f32(param)
        tape32, ofs32, tape64, ofs64 = f32(tape32), u64(ofs32), f64(tape64), u64(ofs64)
        n = len(dirs)
        out, ev, used = np.zeros((n, 3), np.float32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        self.lib.oracle_scatter_tape(C.c_uint32(n), _p(dirs), _p(points), _p(normals), _p(mtype), _p(rgba), _p(param),
                                     _p(tape32), _p(ofs32), _p(tape64), _p(ofs64), _p(out), _p(ev), _p(used))
        return out, ev, used

    def cast_rays_mt(self, h, rays, times, depth, skip32=0, skip64=0, query=QUERY_BVH_QUIRK, product=PRODUCT_RECURSIVE):
        rays, times = f32(rays), f32(times)
        n = len(rays)
        out = np.zeros((n, 4), np.float32)
        n32, n64 = C.c_float(), C.c_double()
        o32, o64, segs = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64), np.zeros(n, np.uint32)
        self.lib.oracle_cast_rays_mt(h, C.c_uint32(n), _p(rays), _p(times), C.c_uint32(depth), C.c_uint32(skip32),
                                     C.c_uint32(skip64), C.c_int(query), C.c_int(product), _p(out), C.byref(n32),
                                     C.byref(n64), _p(o32), _p(o64), _p(segs))
        return out, np.float32(n32.value), n64.value, o32, o64, segs

    def cast_rays_tape(self, h, rays, times, depth, tape32, ofs32, tape64, ofs64, query=QUERY_CLOSEST,
                       product=PRODUCT_ITERATIVE, max_log=0):
        rays, times = f32(rays), f32(times)
        tape32, ofs32, tape64, ofs64 = f32(tape32), u64(ofs32), f64(tape64), u64(ofs64)
        n = len(rays)
        out, flags = np.zeros((n, 4), np.float32), np.zeros(n, np.uint32)
        log = np.zeros((n, max_log, 12), np.float32) if max_log else None
        cnt = np.zeros(n, np.uint32)
        self.lib.oracle_cast_rays_tape(h, C.c_uint32(n), _p(rays), _p(times), C.c_uint32(depth), _p(tape32), _p(ofs32),
                                       _p(tape64), _p(ofs64), C.c_int(query), C.c_int(product), _p(out), _p(flags),
                                       _p(log), C.c_uint32(max_log), _p(cnt))
        return out, flags, log, cnt

    # ---- renders
    def render_rect_mt(self, h, w, hgt, spp, depth, rect, skip32=0, skip64=0, query=QUERY_BVH_QUIRK, product=PRODUCT_RECURSIVE):
        out = np.zeros((hgt, w, 4), np.float32)
        x0, y0, x1, y1 = rect
        self.lib.oracle_render_rect_mt(h, C.c_uint32(w), C.c_uint32(hgt), C.c_uint32(spp), C.c_uint32(depth),
                                       C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(x1), C.c_uint32(y1),
                                       C.c_uint32(skip32), C.c_uint32(skip64), C.c_int(query), C.c_int(product), _p(out))
        return out

    def render_philox(self, h, w, hgt, s0, s1, depth, seed=0, query=QUERY_CLOSEST, product=PRODUCT_ITERATIVE, nthreads=None):
        nthreads = nthreads or min(16, os.cpu_count() or 1)
        out = np.zeros((hgt, w, 4), np.float32)
        seg = self.lib.oracle_render_philox(h, C.c_uint32(w), C.c_uint32(hgt), C.c_uint32(s0), C.c_uint32(s1),
                                            C.c_uint32(depth), C.c_uint64(seed), C.c_int(query), C.c_int(product),
                                            C.c_uint32(nthreads), _p(out))
        return out, int(seg)

    def samples_philox(self, h, w, hgt, pixels, samples, depth, seed=0, query=QUERY_CLOSEST, product=PRODUCT_ITERATIVE):
        pixels, samples = u32(pixels), u32(samples)
        n = len(pixels)
        out, segs = np.zeros((n, 4), np.float32), np.zeros(n, np.uint32)
        self.lib.oracle_samples_philox(h, C.c_uint32(w), C.c_uint32(hgt), C.c_uint32(n), _p(pixels), _p(samples),
                                       C.c_uint32(depth), C.c_uint64(seed), C.c_int(query), C.c_int(product), _p(out), _p(segs))
        return out, segs

    def time_render_philox(self, h, w, hgt, s0, s1, depth, seed, nthreads):
        out = np.zeros((hgt, w, 4), np.float32)
        seg = C.c_uint64()
        secs = self.lib.oracle_time_render_philox(h, C.c_uint32(w), C.c_uint32(hgt), C.c_uint32(s0), C.c_uint32(s1),
                                                  C.c_uint32(depth), C.c_uint64(seed), C.c_uint32(nthreads), _p(out), C.byref(seg))
        return secs, out, seg.value

    # ---- resolve / BMP
    def resolve(self, sum_rgba, spp):
        s = f32(sum_rgba)
        out = np.zeros_like(s)
        self.lib.oracle_resolve(C.c_uint32(s.size // 4), _p(s), C.c_uint32(spp), _p(out))
        return out

    def pack_bgra(self, pixels):
        p = f32(pixels)
        out = np.zeros(p.size // 4, np.uint32)
        self.lib.oracle_pack_bgra(C.c_uint32(p.size // 4), _p(p), _p(out))
        return out

    def bmp_header(self, w, h):
        out = np.zeros(58, np.uint8)
        self.lib.oracle_bmp_header(C.c_uint32(w), C.c_uint32(h), _p(out))
        return out


class Ref:
    """The compiled reference.  Ref.available() is False on a box where it was neither built nor shipped."""

    @staticmethod
    def _path(flavour):
        return REF_SO if flavour is None else REF_SO[:-3] + "_" + flavour + ".so"

    @staticmethod
    def available(flavour=None):
        """flavour None: the -O2 build every test uses.  "o3": the same sources at -O3 -march=x86-64-v3 (bench.py's courtesy
        CPU figure); only offered on a CPU that has AVX2, FMA and BMI2 (the library would fault elsewhere)."""
        if not os.path.exists(Ref._path(flavour)):
            return False
        if flavour == "o3":
            try:
                flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
            except (OSError, StopIteration):
                return False
            return all(f in flags for f in ("avx2", "fma", "bmi2", "movbe", "abm"))
        return True

    def __init__(self, path=None, flavour=None):
        self.lib = L = C.CDLL(path or Ref._path(flavour))
        L.ref_scene_new.restype = C.c_void_p
        L.ref_scene_builtin.restype = C.c_void_p
        L.ref_reflectance.restype = C.c_double
        L.ref_reflectance.argtypes = [C.c_double, C.c_double]
        L.ref_render.restype = C.c_double
        L.ref_scene_build_bvh.restype = C.c_uint32

    def sizeofs(self):
        out = np.zeros(7, np.uint32)
        self.lib.ref_sizeofs(_p(out))
        return dict(zip(("Material", "SphereObject", "PlaneObject", "World", "Camera", "BVH", "Ray"), out.tolist()))

    def random_f32(self, skip, n):
        out = np.zeros(n, np.float32)
        self.lib.ref_random_f32(C.c_uint32(skip), C.c_uint32(n), _p(out))
        return out

    def random_f64(self, skip, n):
        out = np.zeros(n, np.float64)
        self.lib.ref_random_f64(C.c_uint32(skip), C.c_uint32(n), _p(out))
        return out

    def _prim(self, fn, rays, other, out):
        rays, other = f32(rays), f32(other)
        getattr(self.lib, fn)(C.c_uint32(len(out)), _p(rays), _p(other), _p(out))
        return out

    def intersect_spheres(self, rays, spheres):
        return self._prim("ref_intersect_spheres", rays, spheres, np.zeros(len(rays), np.float32))

    def intersect_planes(self, rays, planes):
        return self._prim("ref_intersect_planes", rays, planes, np.zeros(len(rays), np.float32))

    def hit_boxes(self, rays, boxes):
        return self._prim("ref_hit_boxes", rays, boxes, np.zeros(len(rays), np.uint8))

    def reflect(self, dirs, normals):
        return self._prim("ref_reflect", dirs, normals, np.zeros((len(dirs), 3), np.float32))

    def normalize(self, v):
        v, out = f32(v), np.zeros((len(v), 3), np.float32)
        self.lib.ref_normalize(C.c_uint32(len(v)), _p(v), _p(out))
        return out

    def reflectance(self, c, r):
        return self.lib.ref_reflectance(float(c), float(r))

    def _samp(self, fn, skip, n, dim):
        out, nxt = np.zeros((n, dim), np.float32), C.c_float()
        getattr(self.lib, fn)(C.c_uint32(skip), C.c_uint32(n), _p(out), C.byref(nxt))
        return out, np.float32(nxt.value)

    def unit_sphere_points(self, skip, n):
        return self._samp("ref_unit_sphere_points", skip, n, 3)

    def unit_vectors(self, skip, n):
        return self._samp("ref_unit_vectors", skip, n, 3)

    def unit_circle_points(self, skip, n):
        return self._samp("ref_unit_circle_points", skip, n, 2)

    def get_rays(self, camera13, skip, uv):
        cam, uv = f32(camera13), f32(uv)
        out, nxt = np.zeros((len(uv), 6), np.float32), C.c_float()
        self.lib.ref_get_rays(_p(cam), C.c_uint32(skip), C.c_uint32(len(uv)), _p(uv), _p(out), C.byref(nxt))
        return out, np.float32(nxt.value)

    def camera_make(self, pos, fov_deg, aspect, target=None, up=None, lens=None):
        out = np.zeros(13, np.float32)
        pos = f32(pos)
        target = f32(target) if target is not None else None
        up = f32(up) if up is not None else None
        ap, fo = (lens if lens else (0.0, 0.0))
        self.lib.ref_camera_make(_p(pos), C.c_float(fov_deg), C.c_float(aspect), _p(target), _p(up),
                                 C.c_float(ap), C.c_float(fo), C.c_int(1 if lens else 0), _p(out))
        return out

    # ---- scenes
    def scene_builtin(self, scene_id, aspect):
        return C.c_void_p(self.lib.ref_scene_builtin(C.c_int(scene_id), C.c_float(aspect)))

    def scene_from_arrays(self, sa: SceneArrays):
        h = C.c_void_p(self.lib.ref_scene_new())
        for i in range(sa.n_planes):
            ok = self.lib.ref_scene_add_plane(h, _p(sa.pl_n_d[i]), C.c_uint32(int(sa.pl_type[i])), _p(sa.pl_rgba[i]),
                                              C.c_float(sa.pl_param[i, 0]), C.c_float(sa.pl_param[i, 1]))
            assert ok
        for i in range(sa.n_spheres):
            ok = self.lib.ref_scene_add_sphere(h, _p(sa.pos_r[i]), _p(sa.vel[i]), C.c_uint32(int(sa.mtype[i])),
                                               _p(sa.rgba[i]), C.c_float(sa.param[i, 0]), C.c_float(sa.param[i, 1]))
            assert ok
        self.lib.ref_scene_set_time(h, C.c_float(sa.t0), C.c_float(sa.t1))
        self.lib.ref_scene_set_camera(h, _p(sa.camera))
        return h

    def scene_free(self, h):
        self.lib.ref_scene_free(h)

    def scene_arrays(self, h) -> SceneArrays:
        ns, npl, t = C.c_uint32(), C.c_uint32(), (C.c_float * 2)()
        self.lib.ref_scene_counts(h, C.byref(ns), C.byref(npl), t)
        ns, npl = ns.value, npl.value
        pos_r, vel = np.zeros((ns, 4), np.float32), np.zeros((ns, 3), np.float32)
        mtype, rgba, param = np.zeros(ns, np.uint32), np.zeros((ns, 4), np.float32), np.zeros((ns, 2), np.float32)
        pnd, ptype = np.zeros((npl, 4), np.float32), np.zeros(npl, np.uint32)
        prgba, pparam = np.zeros((npl, 4), np.float32), np.zeros((npl, 2), np.float32)
        self.lib.ref_scene_dump(h, _p(pos_r), _p(vel), _p(mtype), _p(rgba), _p(param), _p(pnd), _p(ptype), _p(prgba), _p(pparam))
        cam = np.zeros(13, np.float32)
        self.lib.ref_scene_get_camera(h, _p(cam))
        return SceneArrays(pos_r, vel, mtype, rgba, param, pnd, ptype, prgba, pparam, t[0], t[1], cam)

    def build_bvh(self, h, skip=0):
        return int(self.lib.ref_scene_build_bvh(h, C.c_uint32(skip)))

    def dump_bvh(self, h, n_nodes):
        boxes, links = np.zeros((n_nodes, 6), np.float32), np.zeros((n_nodes, 3), np.int32)
        self.lib.ref_scene_dump_bvh(h, _p(boxes), _p(links))
        return boxes, links

    def bvh_query(self, h, rays, times):
        rays, times = f32(rays), f32(times)
        t, obj = np.zeros(len(rays), np.float32), np.zeros(len(rays), np.int32)
        self.lib.ref_bvh_query(h, C.c_uint32(len(rays)), _p(rays), _p(times), _p(t), _p(obj))
        return t, obj

    def cast_rays(self, h, rays, times, depth, skip32=0, skip64=0):
        rays, times = f32(rays), f32(times)
        out = np.zeros((len(rays), 4), np.float32)
        n32, n64 = C.c_float(), C.c_double()
        self.lib.ref_cast_rays(h, C.c_uint32(len(rays)), _p(rays), _p(times), C.c_uint32(depth), C.c_uint32(skip32),
                               C.c_uint32(skip64), _p(out), C.byref(n32), C.byref(n64))
        return out, np.float32(n32.value), n64.value

    def render_rect(self, h, w, hgt, spp, depth, rect, skip32=0, skip64=0):
        out = np.zeros((hgt, w, 4), np.float32)
        x0, y0, x1, y1 = rect
        self.lib.ref_render_rect(h, C.c_uint32(w), C.c_uint32(hgt), C.c_uint32(spp), C.c_uint32(depth), C.c_uint32(x0),
                                 C.c_uint32(y0), C.c_uint32(x1), C.c_uint32(y1), C.c_uint32(skip32), C.c_uint32(skip64), _p(out))
        return out

    def render(self, h, w, hgt, spp, depth, nthreads, skip32=0):
        out = np.zeros((hgt, w, 4), np.float32)
        secs = self.lib.ref_render(h, C.c_uint32(w), C.c_uint32(hgt), C.c_uint32(spp), C.c_uint32(depth), C.c_uint32(nthreads), C.c_uint32(skip32), _p(out))
        return secs, out
