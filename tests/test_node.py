"""pt_node_* / pt_hip_render_node (include/pt_hip.h): the node-level entry -- spp split, one RCCL all-reduce inside the
library.  CPU: the split logic.  GPU (one device here): the path end to end with one rank, with and without the
collective forced on, against pt_hip_render."""
import numpy as np
import pytest

import pathtracer_amd as pt
from tests._util import bit_equal


@pytest.mark.parametrize("s0,s1", [(0, 4096), (3, 48), (0, 7), (10, 11), (5, 5), (0, 1000003)])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 16])
def test_spp_slices_partition_the_range(s0, s1, n):
    cuts = [pt.spp_slice(s0, s1, n, g) for g in range(n)]
    assert cuts[0][0] == s0 and cuts[-1][1] == max(s0, s1)
    for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
        assert b == c and a <= b                                   # contiguous, ordered, disjoint
    sizes = [b - a for a, b in cuts]
    assert sum(sizes) == max(0, s1 - s0) and max(sizes) - min(sizes) <= 1
    assert pt.spp_slice(s0, s1, n, n + 5) == cuts[-1]              # a part beyond the last is clamped to it
    assert pt.spp_slice(s0, s1, 0, 0) == (s0, max(s0, s1))         # zero parts = one part


def test_node_argument_errors_without_a_device():
    import ctypes as C
    L = pt.load_library()
    h = C.c_void_p()
    assert L.pt_node_create(0, C.byref(h)) == pt.PT_ERR_INVALID
    assert L.pt_node_create(1, None) == pt.PT_ERR_INVALID
    assert L.pt_node_create(100000, C.byref(h)) == pt.PT_ERR_NO_DEVICE
    assert L.pt_node_render(None, None, None, None) == pt.PT_ERR_INVALID
    assert L.pt_node_gpu_count(None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("force_collective", [False, True])
def test_node_render_equals_the_single_context_render(renderer, monkeypatch, force_collective):
    """One GPU: the node's frame IS pt_hip_render's (same sample range, same kernels), and the 0xAARRGGBB pixels are its
    resolve + pack.  With PT_NODE_FORCE_COLLECTIVE=1 the same frame goes through ncclCommInitAll + ncclAllReduce on a
    one-rank communicator: the RCCL path initialised and run on hardware."""
    if force_collective:
        monkeypatch.setenv("PT_NODE_FORCE_COLLECTIVE", "1")
    w, h, spp, depth = 96, 54, 24, 50
    scene = pt.Scene.builtin(1, w / h)
    renderer.upload(scene)
    want = renderer.render(w, h, 2, 2 + spp, depth, seed=7, flags=pt.FLAG_REPRODUCIBLE)
    with pt.Node(1) as node:
        node.upload(scene)
        got, bgra = node.render(w, h, 2, 2 + spp, depth, seed=7, flags=pt.FLAG_REPRODUCIBLE, want_bgra=True)
        st = node.stats(0)
        assert st["samples"] == w * h * spp and st["kernel"] == 3
        assert bit_equal(got, want)
        assert (bgra.reshape(-1) == pt.pack_bgra(pt.resolve(want, spp))).all()
        # a second, smaller frame on the same node; default flags (sums agree to summation order)
        got2 = node.render(40, 24, 0, 9, 12, seed=1)
    renderer.upload(pt.Scene.builtin(1, w / h))
    want2 = renderer.render(40, 24, 0, 9, 12, seed=1)
    assert np.abs(got2 - want2).max() <= 2 * 9 * 2.0 ** -24 * max(1.0, float(np.abs(want2).max()))


@pytest.mark.gpu
def test_render_node_one_call_and_too_many_gpus(renderer):
    import ctypes as C
    L = pt.load_library()
    w, h = 48, 27
    scene = pt.Scene.builtin(3, w / h)
    out = np.zeros((h, w, 4), np.float32)
    p = pt.Params(w, h, 0, 6, 50, pt.FLAG_REPRODUCIBLE, 3)
    assert L.pt_hip_render_node(C.byref(scene.desc), C.byref(p), 1, out.ctypes.data_as(C.c_void_p)) == pt.PT_OK
    renderer.upload(scene)
    assert bit_equal(out, renderer.render(w, h, 0, 6, 50, seed=3, flags=pt.FLAG_REPRODUCIBLE))
    assert L.pt_hip_render_node(C.byref(scene.desc), C.byref(p), pt.device_count() + 1, out.ctypes.data_as(C.c_void_p)) == pt.PT_ERR_NO_DEVICE
