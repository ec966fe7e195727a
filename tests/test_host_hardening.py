"""Host-side hardening (VERDICT r2 missing 3-4): the sanitizer build of the host passes, fault
injection at the C-ABI, and "one tree build, N commits".  CPU only."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "native", "build")


def _make(target):
    r = subprocess.run(["make", "-C", ROOT, target], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_host_passes_under_asan_ubsan(tmp_path):
    """pt_world.cpp + pt_bvh.h (scene builders, scene-file parser on malformed files, both tree builders, the widening with
    leaves of up to 8 spheres) under -fsanitize=address,undefined."""
    _make("asan")
    # (the sanitizer runtime is linked statically and told not to insist on being the first library of the process: the
    # environment's own preloads stay as they are)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(BUILD, "host_sanitize"), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "all checks held" in r.stderr


def test_allocation_failure_returns_a_status():
    """Every allocation of pt_scene_builtin / pt_scene_prepare / pt_reference_bvh / pt_bvh_selfcheck made to fail in turn:
    a status or NULL comes back, the process lives (include/pt_hip.h: nothing throws across the boundary)."""
    _make("oom")
    r = subprocess.run([os.path.join(BUILD, "host_oom")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "host_oom: ok" in r.stdout and "returned PT_ERR_NO_MEMORY, 0 another status" in r.stdout


@pytest.mark.parametrize("limit", ["0", "3"])
def test_thread_spawn_failure_degrades_to_fewer_threads(limit):
    """A worker thread that cannot start (the process limit of a GPU box) is not std::terminate: the builder runs with
    what it gets, down to the calling thread alone, and builds the same tree."""
    _make("oom")
    env = dict(os.environ, PT_TEST_SPAWN_LIMIT=limit, PT_BUILD_THREADS="8")
    r = subprocess.run([os.path.join(BUILD, "host_oom")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert f"PT_TEST_SPAWN_LIMIT={limit}, PT_BUILD_THREADS=8): status 0, 23177 nodes, 70000 spheres in leaves" in r.stdout


def test_prepare_builds_the_tree_once():
    """pt_scene_prepare is the only step that runs the BVH builder (pt_node_upload_scene: one prepare, N commits)."""
    scene = pt.Scene.builtin(2, 16 / 9)
    before = pt.tree_build_count()
    with scene.prepare() as p:
        assert p.h
        assert pt.tree_build_count() == before + 1
    with scene.prepare():
        pass
    assert pt.tree_build_count() == before + 2


def test_prepare_rejects_what_upload_rejects():
    scene = pt.Scene.builtin(4, 1.0)
    scene.desc.n_planes = 65  # more planes than the reference's World holds (src/render_world.h:88)
    with pytest.raises(pt.PtError) as e:
        scene.prepare()
    assert e.value.status == pt.PT_ERR_UNSUPPORTED
    L = pt.load_library()
    assert L.pt_scene_prepare(None, None) == pt.PT_ERR_INVALID
    assert L.pt_hip_commit_scene(None, None) == pt.PT_ERR_INVALID
    assert L.pt_node_snapshot(None, None, None, None) == pt.PT_ERR_INVALID
    assert L.pt_node_reset(None) == pt.PT_ERR_INVALID


def test_selfcheck_with_big_leaves(monkeypatch):
    """PT_LEAF_MAX 5..8 used to overflow the 4-slot wide node of an over-full leaf."""
    monkeypatch.setenv("PT_LEAF_MAX", "8")
    for sid in (1, 2, 3):
        st, c = pt.Scene.builtin(sid, 16 / 9).bvh_selfcheck()
        assert st == pt.PT_OK, c
        assert c["largest_leaf"] <= 8 and c["bad_references"] == 0
