"""CPU: the product's host side and its C-ABI surface (no GPU needed): scene builders vs the reference's scenes,
resolve / pack / BMP vs the oracle and the reference's own output files, exported symbols vs include/pt_hip.h."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pathtracer_amd as pt
from oracle import pyoracle as po
from tests._util import bit_equal, to_scene_arrays

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_exactly_the_header(golden_dir):
    header = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|uint32_t|uint64_t|const char\*|pt_scene\*|pt_ctx\*)\s+(pt_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared == set(pt.EXPORTS), declared ^ set(pt.EXPORTS)
    L = pt.load_library()
    for sym in pt.EXPORTS:
        assert hasattr(L, sym), sym
    # every replaced reference interface is cited (file:line) next to its declaration
    assert len(re.findall(r"src/[a-z_]+\.(?:cpp|h):\d+", header)) >= 25
    # ctypes mirrors of the structs have the C sizes
    assert C.sizeof(pt.Params) == 32 and C.sizeof(pt.Stats) == 88 and C.sizeof(pt.SceneDesc) == 152


def test_status_codes_without_gpu():
    L = pt.load_library()
    assert L.pt_hip_strerror(0) == b"ok" and b"invalid" in L.pt_hip_strerror(pt.PT_ERR_INVALID)
    h = C.c_void_p()
    assert L.pt_hip_create(-1, C.byref(h)) == pt.PT_ERR_NO_DEVICE
    assert L.pt_hip_create(0, None) == pt.PT_ERR_INVALID
    if pt.device_count() == 0:
        assert L.pt_hip_create(0, C.byref(h)) == pt.PT_ERR_NO_DEVICE   # loud failure, no CPU fallback
        with pytest.raises(pt.PtError):
            pt.Renderer(0)
    assert L.pt_resolve(None, 1, 1, None) == pt.PT_ERR_INVALID
    assert L.pt_bmp_header(0, 4, None) == pt.PT_ERR_INVALID
    assert L.pt_scene_builtin(99, 1.0, 0) is None


@pytest.mark.parametrize("sid", [1, 2, 3])
def test_scene_builders_match_reference(golden_dir, sid):
    z = np.load(os.path.join(golden_dir, "scenes.npz"))
    want = po.SceneArrays.from_npz(z, f"s{sid}_")
    got = to_scene_arrays(pt.Scene.builtin(sid, 16 / 9))
    for k in ("pos_r", "vel", "rgba", "param", "pl_n_d", "pl_rgba", "pl_param", "camera"):
        assert bit_equal(getattr(got, k), getattr(want, k)), (sid, k)
    assert (got.mtype == want.mtype).all() and (got.pl_type == want.pl_type).all()
    assert (got.t0, got.t1) == (want.t0, want.t1)


def test_scene_1_facts_from_the_survey():
    s = pt.Scene.builtin(1, 800 / 450)
    assert s.n_spheres == 259 and s.n_planes == 1
    assert np.bincount(s.mtype, minlength=4).tolist() == [0, 222, 2, 35]
    assert np.allclose(s.pos_r[0], [-25.47893, 0.55, -27.14308, 0.6811584], atol=1e-5)
    assert abs(float(np.dot(s.camera[3:6], s.camera[6:9])) + 0.134) < 1e-3          # sheared image plane (8a-2)
    assert s.vel is None
    assert pt.Scene.builtin(3, 1.0).vel is not None


def test_build_defined_scenes():
    m = pt.Scene.builtin(4, 16 / 9)     # C2: materials + DOF
    assert m.n_spheres == 3 and m.n_planes == 1 and sorted(m.mtype.tolist()) == [1, 2, 3]
    assert m.camera[12] == np.float32(0.1) and m.camera[11] == np.float32(8.06)
    a, b = pt.Scene.builtin(5, 16 / 9, 5000), pt.Scene.builtin(5, 16 / 9, 5000)   # C5 generator: deterministic
    assert a.n_spheres == 5000 and bit_equal(a.pos_r, b.pos_r)
    frac = np.bincount(a.mtype, minlength=4) / 5000
    assert abs(frac[1] - 0.5) < 0.03 and abs(frac[2] - 0.4) < 0.03 and abs(frac[3] - 0.1) < 0.02
    assert (a.pos_r[:, 1] - a.pos_r[:, 3] >= -1e-6).all()                           # above the ground plane


def test_resolve_pack_bmp_against_oracle_and_reference_files(oracle, golden_dir, tmp_path):
    rng = np.random.default_rng(2)
    w, h, spp = 37, 11, 13
    s = (rng.uniform(-0.1, 1.3, size=(h, w, 4)) * spp).astype(np.float32)
    px = pt.resolve(s, spp)
    assert bit_equal(px, oracle.resolve(s, spp))
    assert (pt.pack_bgra(px) == oracle.pack_bgra(px)).all()
    z = np.load(os.path.join(golden_dir, "scenes.npz"))
    for head, (bw, bh) in zip(z["bmp_headers"], z["bmp_dims"]):
        if bh < 0:
            assert (pt.bmp_header(int(bw), int(-bh)) == head).all()
    f = tmp_path / "out.bmp"
    pt.bmp_write(str(f), px, w, h)
    data = np.fromfile(f, np.uint8)
    assert len(data) == 58 + 4 * w * h
    assert (data[:58] == oracle.bmp_header(w, h)).all()
    assert (data[58:].view(np.uint32) == oracle.pack_bgra(px)).all()
    with pytest.raises(pt.PtError):
        pt.bmp_write(str(tmp_path / "no_such_dir" / "x.bmp"), px, w, h)


def test_product_never_touches_the_oracle():
    """The product path must not import, link, call or read anything under oracle/ (nor /root/reference)."""
    bad = []
    for base in ("pathtracer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "lib" in dp:
                continue
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".cpp")):
                    text = open(os.path.join(dp, fn), errors="replace").read()
                    # (comments may SAY "the oracle does the same"; what must not appear is a path, module or library)
                    if re.search(r"oracle[/.]|pyoracle|libpt_oracle|libptref|/root/reference|pt_oracle", text):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
    import subprocess
    so = pt.LIB_PATH
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed and "ptref" not in needed


@pytest.mark.parametrize("what", ["scene1", "scene2", "scene3_moving", "materials", "random_3", "random_5000", "random_120000", "random_moving_400", "empty"])
def test_bvh_builder_invariants_on_the_host(what):
    """What the closest-hit parity rests on, checked without a GPU (pt_bvh_selfcheck): every sphere sits in exactly one
    leaf and inside that leaf's box (swept over the shutter interval for moving spheres), references stay in range, and
    the 32-byte grid nodes big trees use enclose the float boxes they encode -- also for the multi-threaded build."""
    from tests import _gen
    from tests._util import to_pt_scene
    rng = np.random.default_rng(31)
    if what == "scene1": sc = pt.Scene.builtin(1, 16 / 9)
    elif what == "scene2": sc = pt.Scene.builtin(2, 16 / 9)
    elif what == "scene3_moving": sc = pt.Scene.builtin(3, 16 / 9)
    elif what == "materials": sc = pt.Scene.builtin(4, 16 / 9)
    elif what == "random_3": sc = to_pt_scene(_gen.random_scene(rng, 3, n_planes=1))
    elif what == "random_5000": sc = pt.Scene.builtin(5, 16 / 9, 5000)
    elif what == "random_120000": sc = pt.Scene.builtin(5, 16 / 9, 120000)     # above the builder's threading threshold
    elif what == "random_moving_400": sc = to_pt_scene(_gen.random_scene(rng, 400, n_planes=2, moving=True, extent=14.0))
    else: sc = to_pt_scene(_gen.random_scene(rng, 0, n_planes=2))
    st, c = sc.bvh_selfcheck()
    assert st == pt.PT_OK, c
    assert c["spheres_in_leaves"] == sc.n_spheres and c["largest_leaf"] <= 4
    assert c["spheres_outside_leaf_box"] == c["grid_boxes_not_enclosing"] == c["bad_references"] == 0
    if sc.n_spheres > 4:
        assert c["nodes"] >= (sc.n_spheres + 3) // 4 - 1 and c["depth"] < 64


def test_chunk_schedule_partitions_the_sample_range():
    """The guided schedule the render kernel walks (host-only helper): sizes sum to nspp, respect the minimum (a last
    remainder below 2 x min goes in one piece), never exceed half of what is left, and shrink monotonically."""
    rng = np.random.default_rng(2)
    cases = [(1, 0x8000, 8), (7, 0x8000, 8), (15, 0x8000, 8), (16, 0x8000, 8), (512, 0x8000, 8), (512, 0x1000, 8), (4096, 0, 8), (4096, 123, 1)]
    cases += [(int(rng.integers(1, 100000)), int(rng.integers(0, 0x8001)), int(rng.integers(1, 64))) for _ in range(200)]
    for nspp, ratio, m in cases:
        sizes = pt.chunk_schedule(nspp, ratio, m)
        assert sum(sizes) == nspp and all(s > 0 for s in sizes), (nspp, ratio, m)
        left = nspp
        for i, s in enumerate(sizes):
            if left < 2 * m:
                assert s == left and i == len(sizes) - 1
            else:
                assert m <= s <= max(m, left // 2)
            left -= s
        assert all(a >= b for a, b in zip(sizes[:-2], sizes[1:-1]))          # big first (the last piece may be the remainder)
    assert pt.chunk_schedule(512, 0x8000, 8) == [256, 128, 64, 32, 16, 8, 8]
    assert pt.chunk_schedule(0, 0x8000, 8) == []


def test_scene_files_round_trip_exactly(tmp_path):
    """pt_scene_save / pt_scene_load: every float of every built-in scene survives the text form."""
    for sid, n in ((1, 0), (2, 0), (3, 0), (4, 0), (5, 300)):
        a = pt.Scene.builtin(sid, 16 / 9, n)
        f = tmp_path / f"scene{sid}.txt"
        a.save(str(f))
        b = pt.Scene.load(str(f))
        assert a.n_spheres == b.n_spheres and a.n_planes == b.n_planes
        for name in ("pos_r", "mtype", "rgba", "param", "pl_n_d", "pl_type", "pl_rgba", "pl_param", "camera"):
            x, y = getattr(a, name), getattr(b, name)
            assert x.dtype == y.dtype and np.array_equal(x.view(np.uint32), y.view(np.uint32)), (sid, name)
        assert (a.vel is None) == (b.vel is None) and (a.vel is None or bit_equal(a.vel, b.vel)), sid
        assert (a.t0, a.t1) == (b.t0, b.t1)
    bad = tmp_path / "bad.txt"
    for text in ("", "sphere 0 0 0 1 0 0 0 1 1 1 1 1 0 0\n", "pt_scene 2\n", "pt_scene 1\ncamera 1 2 3\n", "pt_scene 1\ntime 0 1\n",
                 "pt_scene 1\ncamera 0 0 0 0 0 -1 0 1 0 1 1 1 0\nsphere 0 0 0 1 0 0 0 9 1 1 1 1 0 0\n", "pt_scene 1\nbogus 1\n"):
        bad.write_text(text)
        with pytest.raises(pt.PtError):
            pt.Scene.load(str(bad))
    with pytest.raises(pt.PtError):
        pt.Scene.load(str(tmp_path / "missing.txt"))


def test_both_builds_keep_the_two_flags_the_kernels_depend_on():
    """-ffp-contract=off is parity (value-carrying arithmetic rounds like the reference's); -fno-slp-vectorize is 7 % of the
    bench frame (the compiler otherwise pairs scalar FP32 work into half-rate packed instructions).  Makefile and
    CMakeLists.txt must agree on both."""
    for name, pattern in (("Makefile", r"^HIPFLAGS\s*:=\s*(.*)$"), ("CMakeLists.txt", r"^set\(PT_HIPFLAGS\s+(.*)\)$")):
        text = open(os.path.join(ROOT, name)).read()
        m = re.search(pattern, text, re.M)
        assert m, name
        flags = m.group(1).split()
        assert "-ffp-contract=off" in flags and "-fno-slp-vectorize" in flags and "-O3" in flags, (name, flags)
