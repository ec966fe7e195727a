"""pathtracer_amd -- Python face of libpt_hip.so (include/pt_hip.h).

Thin ctypes plumbing for tests, bench.py and the multi-GPU driver: every call goes through the C-ABI, nothing is
computed in Python and there is NO CPU fallback -- if the HIP library is missing or no MI355X is visible the calls
raise.  Device buffers handed in/out are raw addresses (ints), e.g. ``torch.Tensor.data_ptr()``.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpt_hip.so")

PT_OK = 0
PT_ERR_INVALID, PT_ERR_NO_DEVICE, PT_ERR_HIP, PT_ERR_NO_SCENE, PT_ERR_UNSUPPORTED, PT_ERR_IO = -1, -2, -3, -4, -5, -6
PT_ERR_NO_MEMORY, PT_ERR_INTERNAL = -7, -8
MAT_NONE, MAT_DIFFUSE, MAT_METAL, MAT_DIALECTRIC = 0, 1, 2, 3
FLAG_STRICT_SEARCH, FLAG_FORCE_BVH, FLAG_FORCE_LDS, FLAG_BRUTE_FORCE, FLAG_REPRODUCIBLE, FLAG_REFERENCE_QUERY = 1, 2, 4, 8, 16, 32
FLAG_ACCUMULATE = 64
REFERENCE_MAX_SPHERES = 4192
PRIM_MISS = -(2 ** 31)

# Every symbol include/pt_hip.h declares (tests/test_abi.py checks the library exports exactly these).
EXPORTS = (
    "pt_hip_device_count", "pt_hip_create", "pt_hip_destroy", "pt_hip_strerror", "pt_hip_last_error",
    "pt_hip_upload_scene", "pt_scene_prepare", "pt_hip_commit_scene", "pt_prepared_free", "pt_tree_build_count", "pt_hip_build_reference_tree", "pt_reference_bvh", "pt_hip_render", "pt_hip_render_device", "pt_hip_render_device_static", "pt_hip_last_stats", "pt_hip_debug_counters", "pt_chunk_schedule", "pt_bvh_selfcheck", "pt_hip_resolve_device",
    "pt_resolve", "pt_pack_bgra", "pt_bmp_write", "pt_bmp_write_bgra", "pt_bmp_header",
    "pt_scene_builtin", "pt_scene_get_desc", "pt_scene_free", "pt_scene_save", "pt_scene_load",
    "pt_hip_probe_primitives", "pt_hip_probe_hits", "pt_hip_probe_get_rays", "pt_hip_probe_scatter", "pt_hip_probe_paths",
    "pt_hip_probe_samples", "pt_hip_probe_philox",
    "pt_node_create", "pt_node_upload_scene", "pt_node_render", "pt_node_destroy", "pt_node_last_error", "pt_node_gpu_count", "pt_node_ctx", "pt_node_snapshot", "pt_node_reset", "pt_node_reserve",
    "pt_hip_render_node", "pt_spp_slice",
)


class PtError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: status {status} ({_strerror(status)}){' -- ' + detail if detail else ''}")


class SceneDesc(C.Structure):  # pt_scene_desc
    _fields_ = [
        ("n_spheres", C.c_uint32), ("sphere_pos_r", C.c_void_p), ("sphere_vel", C.c_void_p),
        ("sphere_mat_type", C.c_void_p), ("sphere_mat_rgba", C.c_void_p), ("sphere_mat_param", C.c_void_p),
        ("n_planes", C.c_uint32), ("plane_n_d", C.c_void_p), ("plane_mat_type", C.c_void_p),
        ("plane_mat_rgba", C.c_void_p), ("plane_mat_param", C.c_void_p),
        ("t0", C.c_float), ("t1", C.c_float), ("camera", C.c_float * 13),
    ]


class Params(C.Structure):  # pt_params
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp_begin", C.c_uint32), ("spp_end", C.c_uint32),
                ("max_depth", C.c_uint32), ("flags", C.c_uint32), ("seed", C.c_uint64)]


class Stats(C.Structure):  # pt_stats
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("sphere_tests", C.c_uint64), ("node_visits", C.c_uint64),
                ("wave_iterations", C.c_uint64), ("phase_cycles", C.c_uint64 * 3),
                ("kernel_ms", C.c_double), ("kernel", C.c_uint32), ("node_bytes", C.c_uint32), ("node_width", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def load_library(path=LIB_PATH):
    """Load libpt_hip.so; raises if it was not built (run `make lib` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP runtime and publish it process-wide (RTLD_GLOBAL).  If torch is going
    # to live in this process it must initialise FIRST, so that libpt_hip.so binds to the same runtime (one runtime,
    # shared streams and device pointers); loaded the other way round torch reports "No HIP GPUs are available".
    if "torch" not in sys.modules and os.environ.get("PT_HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: build it with `make lib` (hipcc --offload-arch=gfx950); "
                                "there is no CPU fallback")
    L = C.CDLL(path)
    L.pt_hip_strerror.restype = C.c_char_p
    L.pt_hip_last_error.restype = C.c_char_p
    L.pt_hip_last_error.argtypes = [C.c_void_p]
    L.pt_scene_builtin.restype = C.c_void_p
    L.pt_scene_builtin.argtypes = [C.c_int, C.c_float, C.c_uint32]
    L.pt_scene_get_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.pt_scene_free.argtypes = [C.c_void_p]
    L.pt_scene_save.argtypes = [C.POINTER(SceneDesc), C.c_char_p]
    L.pt_scene_load.argtypes = [C.c_char_p]
    L.pt_scene_load.restype = C.c_void_p
    L.pt_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.pt_hip_destroy.argtypes = [C.c_void_p]
    L.pt_hip_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.pt_scene_prepare.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
    L.pt_hip_commit_scene.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_prepared_free.argtypes = [C.c_void_p]
    L.pt_prepared_free.restype = None
    L.pt_tree_build_count.restype = C.c_uint64
    L.pt_hip_build_reference_tree.argtypes = [C.c_void_p, C.c_uint32]
    L.pt_reference_bvh.argtypes = [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.pt_reference_bvh.restype = C.c_uint32
    L.pt_hip_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p]
    L.pt_hip_render_device.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]
    L.pt_hip_render_device_static.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]
    L.pt_hip_last_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.pt_hip_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 16)]
    L.pt_chunk_schedule.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
    L.pt_chunk_schedule.restype = C.c_uint32
    L.pt_bvh_selfcheck.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.pt_bvh_selfcheck.restype = C.c_int
    L.pt_hip_resolve_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pt_node_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.pt_node_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.pt_node_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p]
    L.pt_node_snapshot.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.pt_node_reset.argtypes = [C.c_void_p]
    L.pt_node_reserve.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.pt_node_destroy.argtypes = [C.c_void_p]
    L.pt_node_last_error.argtypes = [C.c_void_p]
    L.pt_node_last_error.restype = C.c_char_p
    L.pt_node_gpu_count.argtypes = [C.c_void_p]
    L.pt_node_ctx.argtypes = [C.c_void_p, C.c_int]
    L.pt_node_ctx.restype = C.c_void_p
    L.pt_hip_render_node.argtypes = [C.POINTER(SceneDesc), C.POINTER(Params), C.c_int, C.c_void_p]
    L.pt_spp_slice.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pt_spp_slice.restype = None
    _lib = L
    return L


def _strerror(status):
    try:
        return load_library().pt_hip_strerror(C.c_int(status)).decode()
    except Exception:  # noqa: BLE001 - only used to decorate another error
        return "?"


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Scene:
    """Flat scene arrays + the pt_scene_desc that points into them (keeps the arrays alive)."""

    def __init__(self, pos_r, vel, mtype, rgba, param, pl_n_d, pl_type, pl_rgba, pl_param, t0, t1, camera13):
        self.pos_r = _f32(pos_r).reshape(-1, 4)
        n = len(self.pos_r)
        self.vel = None if vel is None else _f32(vel).reshape(-1, 3)
        self.mtype, self.rgba, self.param = _u32(mtype).reshape(-1), _f32(rgba).reshape(-1, 4), _f32(param).reshape(-1, 2)
        self.pl_n_d, self.pl_type = _f32(pl_n_d).reshape(-1, 4), _u32(pl_type).reshape(-1)
        self.pl_rgba, self.pl_param = _f32(pl_rgba).reshape(-1, 4), _f32(pl_param).reshape(-1, 2)
        self.t0, self.t1, self.camera = float(t0), float(t1), _f32(camera13).reshape(13)
        if not (len(self.mtype) == len(self.rgba) == len(self.param) == n) or (self.vel is not None and len(self.vel) != n):
            raise ValueError("sphere arrays disagree in length")
        if not (len(self.pl_type) == len(self.pl_rgba) == len(self.pl_param) == len(self.pl_n_d)):
            raise ValueError("plane arrays disagree in length")
        d = self.desc = SceneDesc()
        d.n_spheres, d.n_planes = n, len(self.pl_n_d)
        d.sphere_pos_r, d.sphere_vel = self.pos_r.ctypes.data, (None if self.vel is None else self.vel.ctypes.data)
        d.sphere_mat_type, d.sphere_mat_rgba, d.sphere_mat_param = self.mtype.ctypes.data, self.rgba.ctypes.data, self.param.ctypes.data
        d.plane_n_d, d.plane_mat_type = self.pl_n_d.ctypes.data, self.pl_type.ctypes.data
        d.plane_mat_rgba, d.plane_mat_param = self.pl_rgba.ctypes.data, self.pl_param.ctypes.data
        d.t0, d.t1 = self.t0, self.t1
        for i in range(13):
            d.camera[i] = float(self.camera[i])

    @property
    def n_spheres(self):
        return len(self.pos_r)

    @property
    def n_planes(self):
        return len(self.pl_n_d)

    @staticmethod
    def _from_handle(h, what):
        L = load_library()
        if not h:
            raise PtError(PT_ERR_INVALID, what)
        try:
            d = SceneDesc()
            st = L.pt_scene_get_desc(h, C.byref(d))
            if st != PT_OK:
                raise PtError(st, "pt_scene_get_desc")

            def arr(ptr, n, dt):
                if not ptr or n == 0:
                    return np.zeros(n, dt)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float if dt == np.float32 else C.c_uint32)), (n,)).astype(dt).copy()

            ns, npl = d.n_spheres, d.n_planes
            return Scene(arr(d.sphere_pos_r, 4 * ns, np.float32), arr(d.sphere_vel, 3 * ns, np.float32) if d.sphere_vel else None,
                         arr(d.sphere_mat_type, ns, np.uint32), arr(d.sphere_mat_rgba, 4 * ns, np.float32),
                         arr(d.sphere_mat_param, 2 * ns, np.float32), arr(d.plane_n_d, 4 * npl, np.float32),
                         arr(d.plane_mat_type, npl, np.uint32), arr(d.plane_mat_rgba, 4 * npl, np.float32),
                         arr(d.plane_mat_param, 2 * npl, np.float32), d.t0, d.t1, np.array(list(d.camera), np.float32))
        finally:
            L.pt_scene_free(h)

    @staticmethod
    def builtin(scene_id, aspect, n_spheres=0):
        """init_test_scene_{1,2,3} / materials (4) / random spheres (5) via the library's host-side builders."""
        return Scene._from_handle(load_library().pt_scene_builtin(scene_id, aspect, n_spheres), f"pt_scene_builtin({scene_id})")

    @staticmethod
    def load(path):
        """A scene file written by save() / pt_scene_save (text; format in pathtracer_amd/host/pt_world.cpp)."""
        return Scene._from_handle(load_library().pt_scene_load(os.fsencode(path)), f"pt_scene_load({path})")

    def bvh_selfcheck(self):
        """Host-only: build this scene's BVH in both node formats and check its invariants (include/pt_hip.h).
        Returns (status, dict of the eight counters)."""
        out = (C.c_uint64 * 8)()
        st = load_library().pt_bvh_selfcheck(C.byref(self.desc), out)
        keys = ("nodes", "depth", "leaves", "largest_leaf", "spheres_in_leaves", "spheres_outside_leaf_box", "grid_boxes_not_enclosing", "bad_references")
        return st, dict(zip(keys, (int(v) for v in out)))

    def reference_bvh(self, rng_skip=0):
        """Host-only: the reference-topology tree (build_bvh_tree's recipe, include/pt_hip.h: pt_reference_bvh).
        -> (boxes (nn, 6), links (nn, 3), order (n,), depth); nodes in the reference's allocation order, root last."""
        n = self.n_spheres
        nn = max(0, 2 * n - 1)
        boxes, links, order = np.zeros((nn, 6), np.float32), np.zeros((nn, 3), np.int32), np.zeros(n, np.uint32)
        depth = C.c_uint32(0)
        got = load_library().pt_reference_bvh(C.byref(self.desc), rng_skip, nn, _ptr(boxes), _ptr(links), _ptr(order), C.byref(depth))
        if got != nn:
            raise PtError(PT_ERR_UNSUPPORTED if n > REFERENCE_MAX_SPHERES else PT_ERR_INVALID, "pt_reference_bvh")
        return boxes, links, order, int(depth.value)

    def prepare(self):
        """Host-only half of an upload (pt_scene_prepare): -> Prepared, to be committed to any number of Renderers."""
        return Prepared(self)

    def save(self, path):
        st = load_library().pt_scene_save(C.byref(self.desc), os.fsencode(path))
        if st != PT_OK:
            raise PtError(st, f"pt_scene_save({path})")


class Prepared:
    """pt_prepared: a scene laid out for the device, tree built, still on the host."""

    def __init__(self, scene: Scene):
        self.L = load_library()
        h = C.c_void_p()
        st = self.L.pt_scene_prepare(C.byref(scene.desc), C.byref(h))
        if st != PT_OK:
            raise PtError(st, "pt_scene_prepare")
        self.h, self.scene = h, scene

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_prepared_free(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def tree_build_count():
    """Runs of the BVH builder in this process so far (pt_tree_build_count)."""
    return int(load_library().pt_tree_build_count())


def device_count():
    return int(load_library().pt_hip_device_count())


class Renderer:
    """One pt_ctx: one GPU, one host thread at a time."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        st = self.L.pt_hip_create(device, C.byref(h))
        if st != PT_OK:
            raise PtError(st, f"pt_hip_create(device={device})")
        self.h = h
        self.scene = None

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_hip_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _check(self, st, what):
        if st != PT_OK:
            raise PtError(st, what, self.L.pt_hip_last_error(self.h).decode() if st in (PT_ERR_HIP, PT_ERR_UNSUPPORTED, PT_ERR_NO_MEMORY, PT_ERR_INTERNAL) else "")

    def upload(self, scene: Scene):
        self._check(self.L.pt_hip_upload_scene(self.h, C.byref(scene.desc)), "pt_hip_upload_scene")
        self.scene = scene

    def commit(self, prepared: Prepared):
        self._check(self.L.pt_hip_commit_scene(self.h, prepared.h), "pt_hip_commit_scene")
        self.scene = prepared.scene

    def build_reference_tree(self, rng_skip=0):
        """The reference-topology tree PT_FLAG_REFERENCE_QUERY traverses (rng_skip: draws taken from the main thread's
        MT19937 stream before build_bvh_tree)."""
        self._check(self.L.pt_hip_build_reference_tree(self.h, rng_skip), "pt_hip_build_reference_tree")

    @staticmethod
    def params(width, height, spp_begin, spp_end, max_depth, seed=0, flags=0):
        return Params(width, height, spp_begin, spp_end, max_depth, flags, seed)

    def render(self, width, height, spp_begin, spp_end, max_depth, seed=0, flags=0):
        """-> (height, width, 4) float32 per-pixel SUM over samples [spp_begin, spp_end)."""
        out = np.empty((height, width, 4), np.float32)
        p = self.params(width, height, spp_begin, spp_end, max_depth, seed, flags)
        self._check(self.L.pt_hip_render(self.h, C.byref(p), _ptr(out)), "pt_hip_render")
        return out

    def render_device(self, params: Params, accum_dev_ptr: int, stream: int = 0):
        self._check(self.L.pt_hip_render_device(self.h, C.byref(params), C.c_void_p(accum_dev_ptr), C.c_void_p(stream)), "pt_hip_render_device")

    def render_device_static(self, params: Params, accum_dev_ptr: int, stream: int = 0):
        """pt_hip_render_device through the entry point that allocates nothing (scratch in a static device array)."""
        self._check(self.L.pt_hip_render_device_static(self.h, C.byref(params), C.c_void_p(accum_dev_ptr), C.c_void_p(stream)), "pt_hip_render_device_static")

    def resolve_device(self, accum_dev_ptr, n_pixels, spp_total, pixels_dev_ptr=0, bgra_dev_ptr=0, stream=0):
        self._check(self.L.pt_hip_resolve_device(self.h, C.c_void_p(accum_dev_ptr), n_pixels, spp_total,
                                                 C.c_void_p(pixels_dev_ptr) if pixels_dev_ptr else None,
                                                 C.c_void_p(bgra_dev_ptr) if bgra_dev_ptr else None, C.c_void_p(stream)), "pt_hip_resolve_device")

    def stats(self):
        s = Stats()
        self._check(self.L.pt_hip_last_stats(self.h, C.byref(s)), "pt_hip_last_stats")
        d = {k: getattr(s, k) for k, _ in Stats._fields_ if k not in ("phase_cycles", "reserved")}
        d["phase_cycles"] = list(s.phase_cycles)
        return d

    def debug_counters(self):
        out = (C.c_uint64 * 16)()
        self._check(self.L.pt_hip_debug_counters(self.h, C.byref(out)), "pt_hip_debug_counters")
        return list(out)

    # ---- parity probes
    def probe_primitives(self, rays, spheres, planes):
        rays, spheres, planes = _f32(rays).reshape(-1, 6), _f32(spheres).reshape(-1, 4), _f32(planes).reshape(-1, 4)
        n = len(rays)
        ts, tp = np.zeros(n, np.float32), np.zeros(n, np.float32)
        self._check(self.L.pt_hip_probe_primitives(self.h, C.c_uint32(n), _ptr(rays), _ptr(spheres), _ptr(planes), _ptr(ts), _ptr(tp)), "pt_hip_probe_primitives")
        return ts, tp

    def probe_hits(self, rays, times, flags=0):
        rays, times = _f32(rays).reshape(-1, 6), _f32(times).reshape(-1)
        n = len(rays)
        t, prim = np.zeros(n, np.float32), np.zeros(n, np.int32)
        pt, nr = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        self._check(self.L.pt_hip_probe_hits(self.h, C.c_uint32(n), _ptr(rays), _ptr(times), C.c_uint32(flags), _ptr(t), _ptr(prim), _ptr(pt), _ptr(nr)), "pt_hip_probe_hits")
        return t, prim, pt, nr

    def probe_get_rays(self, uv, tape, ofs):
        uv, tape, ofs = _f32(uv).reshape(-1, 2), _f32(tape), _u64(ofs)
        out = np.zeros((len(uv), 6), np.float32)
        self._check(self.L.pt_hip_probe_get_rays(self.h, C.c_uint32(len(uv)), _ptr(uv), _ptr(tape), _ptr(ofs), _ptr(out)), "pt_hip_probe_get_rays")
        return out

    def probe_scatter(self, dirs, points, normals, mtype, rgba, param, tape32, ofs32, tape64, ofs64):
        dirs, points, normals = _f32(dirs).reshape(-1, 3), _f32(points).reshape(-1, 3), _f32(normals).reshape(-1, 3)
        mtype, rgba, param = _u32(mtype), _f32(rgba), _f32(param)
        tape32, ofs32, ofs64 = _f32(tape32), _u64(ofs32), _u64(ofs64)
        tape64 = np.ascontiguousarray(tape64, np.float64)
        n = len(dirs)
        out, ev = np.zeros((n, 3), np.float32), np.zeros(n, np.uint32)
        self._check(self.L.pt_hip_probe_scatter(self.h, C.c_uint32(n), _ptr(dirs), _ptr(points), _ptr(normals), _ptr(mtype), _ptr(rgba), _ptr(param),
                                                _ptr(tape32), _ptr(ofs32), _ptr(tape64), _ptr(ofs64), _ptr(out), _ptr(ev)), "pt_hip_probe_scatter")
        return out, ev

    def probe_paths(self, rays, times, max_depth, tape32, ofs32, tape64, ofs64, flags=0):
        rays, times = _f32(rays).reshape(-1, 6), _f32(times).reshape(-1)
        tape32, ofs32, ofs64 = _f32(tape32), _u64(ofs32), _u64(ofs64)
        tape64 = np.ascontiguousarray(tape64, np.float64)
        n = len(rays)
        out, segs = np.zeros((n, 4), np.float32), np.zeros(n, np.uint32)
        self._check(self.L.pt_hip_probe_paths(self.h, C.c_uint32(n), _ptr(rays), _ptr(times), C.c_uint32(max_depth), C.c_uint32(flags),
                                              _ptr(tape32), _ptr(ofs32), _ptr(tape64), _ptr(ofs64), _ptr(out), _ptr(segs)), "pt_hip_probe_paths")
        return out, segs

    def probe_samples(self, width, height, pixels, samples, max_depth, seed=0, flags=0):
        pixels, samples = _u32(pixels), _u32(samples)
        n = len(pixels)
        out, segs = np.zeros((n, 4), np.float32), np.zeros(n, np.uint32)
        p = self.params(width, height, 0, 1, max_depth, seed, flags)
        self._check(self.L.pt_hip_probe_samples(self.h, C.byref(p), C.c_uint32(n), _ptr(pixels), _ptr(samples), _ptr(out), _ptr(segs)), "pt_hip_probe_samples")
        return out, segs

    def probe_philox(self, seed, pixel, sample, n):
        out = np.zeros(n, np.uint32)
        self._check(self.L.pt_hip_probe_philox(self.h, C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(n), _ptr(out)), "pt_hip_probe_philox")
        return out


class Node:
    """pt_node: every GPU of one node behind one handle -- spp split + one RCCL all-reduce inside the library."""

    def __init__(self, n_gpus):
        self.L = load_library()
        h = C.c_void_p()
        st = self.L.pt_node_create(n_gpus, C.byref(h))
        if st != PT_OK:
            raise PtError(st, f"pt_node_create({n_gpus})")
        self.h, self.n = h, n_gpus

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_node_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st, what):
        if st != PT_OK:
            raise PtError(st, what, self.L.pt_node_last_error(self.h).decode())

    def upload(self, scene: Scene):
        self._check(self.L.pt_node_upload_scene(self.h, C.byref(scene.desc)), "pt_node_upload_scene")

    def render(self, width, height, spp_begin, spp_end, max_depth, seed=0, flags=0, want_bgra=False):
        """-> (height, width, 4) f32 sums over the whole sample range [, (height, width) uint32 0xAARRGGBB]."""
        out = np.empty((height, width, 4), np.float32)
        bgra = np.empty((height, width), np.uint32) if want_bgra else None
        p = Params(width, height, spp_begin, spp_end, max_depth, flags, seed)
        self._check(self.L.pt_node_render(self.h, C.byref(p), _ptr(out), _ptr(bgra)), "pt_node_render")
        return (out, bgra) if want_bgra else out

    def accumulate(self, width, height, spp_begin, spp_end, max_depth, seed=0, flags=0):
        """One piece of a progressive render: added to the node's resident sums, nothing read back (asynchronous)."""
        p = Params(width, height, spp_begin, spp_end, max_depth, flags | FLAG_ACCUMULATE, seed)
        self._check(self.L.pt_node_render(self.h, C.byref(p), None, None), "pt_node_render(PT_FLAG_ACCUMULATE)")

    def snapshot(self, width, height, want_sums=True, want_bgra=True):
        """-> (sums or None, bgra or None, spp so far) of the resident running sums (pt_node_snapshot)."""
        out = np.empty((height, width, 4), np.float32) if want_sums else None
        bgra = np.empty((height, width), np.uint32) if want_bgra else None
        spp = C.c_uint64(0)
        self._check(self.L.pt_node_snapshot(self.h, _ptr(out), _ptr(bgra), C.byref(spp)), "pt_node_snapshot")
        return out, bgra, int(spp.value)

    def reset(self):
        self._check(self.L.pt_node_reset(self.h), "pt_node_reset")

    def reserve(self, width, height):
        """Allocate a frame's buffers on every GPU ahead of the first render (the reference allocates its image before its
        timed span, src/main.cpp:385)."""
        self._check(self.L.pt_node_reserve(self.h, width, height), "pt_node_reserve")

    def stats(self, gpu=0):
        s = Stats()
        st = self.L.pt_hip_last_stats(C.c_void_p(self.L.pt_node_ctx(self.h, gpu)), C.byref(s))
        if st != PT_OK:
            raise PtError(st, "pt_hip_last_stats")
        return {k: getattr(s, k) for k, _ in Stats._fields_ if k not in ("phase_cycles", "reserved")}


def spp_slice(spp_begin, spp_end, n_parts, part):
    """Host-only: the sample slice of part `part` of `n_parts` (pt_spp_slice)."""
    a, b = C.c_uint32(), C.c_uint32()
    load_library().pt_spp_slice(spp_begin, spp_end, n_parts, part, C.byref(a), C.byref(b))
    return a.value, b.value


# ---- host-only helpers (no GPU needed) -----------------------------------------------------------------------
def chunk_schedule(nspp, ratio_q16, min_spp):
    """The sample-chunk sizes the render kernel walks (host-only; see include/pt_hip.h)."""
    L = load_library()
    n = L.pt_chunk_schedule(nspp, ratio_q16, min_spp, None, 0)
    out = (C.c_uint32 * max(1, n))()
    L.pt_chunk_schedule(nspp, ratio_q16, min_spp, out, n)
    return list(out)[:n]


def resolve(sum_rgba, spp_total):
    s = _f32(sum_rgba)
    out = np.empty_like(s)
    st = load_library().pt_resolve(_ptr(s), C.c_uint32(s.size // 4), C.c_uint32(spp_total), _ptr(out))
    if st != PT_OK:
        raise PtError(st, "pt_resolve")
    return out


def pack_bgra(pixels_rgba):
    p = _f32(pixels_rgba)
    out = np.empty(p.size // 4, np.uint32)
    st = load_library().pt_pack_bgra(_ptr(p), C.c_uint32(p.size // 4), _ptr(out))
    if st != PT_OK:
        raise PtError(st, "pt_pack_bgra")
    return out


def bmp_header(width, height):
    out = np.zeros(58, np.uint8)
    st = load_library().pt_bmp_header(C.c_uint32(width), C.c_uint32(height), _ptr(out))
    if st != PT_OK:
        raise PtError(st, "pt_bmp_header")
    return out


def bmp_write(path, pixels_rgba, width, height):
    p = _f32(pixels_rgba)
    st = load_library().pt_bmp_write(os.fsencode(path), _ptr(p), C.c_uint32(width), C.c_uint32(height))
    if st != PT_OK:
        raise PtError(st, f"pt_bmp_write({path})")
