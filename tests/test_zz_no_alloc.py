"""GPU: pt_hip_render_device_static -- the entry point that allocates nothing (scratch in a static __device__ array) --
against pt_hip_render_device, through the C-ABI.  Two calls in a row, the second with a smaller input: the array keeps
the first call's contents, so whatever must start at zero has to be cleared by the call itself."""
import numpy as np
import pytest

import pathtracer_amd as pt
from tests._util import bit_equal

pytestmark = pytest.mark.gpu


def _frame(renderer, torch, static, w, h, s0, s1, depth, seed, flags):
    acc = torch.full((h, w, 4), -7.0, dtype=torch.float32, device="cuda:0")     # poisoned: the call must define every pixel
    stream = torch.cuda.current_stream().cuda_stream
    p = renderer.params(w, h, s0, s1, depth, seed=seed, flags=flags)
    (renderer.render_device_static if static else renderer.render_device)(p, acc.data_ptr(), stream)
    torch.cuda.synchronize()
    st = renderer.stats()
    return acc.cpu().numpy(), st


# (scene id, width, height, spp range, depth): a frame cut into several sample chunks (45 spp -> chunks of 8 + a rest:
# partial images in the static array with PT_FLAG_REPRODUCIBLE), then a SMALLER one, then the bigger one again
CALLS = [(1, 88, 56, 3, 48, 12), (1, 40, 24, 0, 20, 12), (3, 64, 36, 0, 45, 50), (1, 88, 56, 3, 48, 12)]


@pytest.mark.parametrize("flags", [pt.FLAG_REPRODUCIBLE, pt.FLAG_REPRODUCIBLE | pt.FLAG_FORCE_BVH, 0, pt.FLAG_BRUTE_FORCE, pt.FLAG_REPRODUCIBLE | pt.FLAG_REFERENCE_QUERY])
def test_static_entry_point_matches_render_device(renderer, flags):
    torch = pytest.importorskip("torch")
    for sid, w, h, s0, s1, depth in CALLS:
        renderer.upload(pt.Scene.builtin(sid, w / h))
        if flags & pt.FLAG_REFERENCE_QUERY:
            renderer.build_reference_tree(0)                                        # (an allocation: before the call that must not allocate)
        want, st_want = _frame(renderer, torch, False, w, h, s0, s1, depth, 5, flags)
        got, st_got = _frame(renderer, torch, True, w, h, s0, s1, depth, 5, flags)
        for k in ("samples", "segments", "kernel", "node_bytes"):
            assert st_got[k] == st_want[k], (k, sid, w, h, flags)
        spp = s1 - s0
        assert np.allclose(got[..., 3], spp, rtol=1e-6), (sid, w, h, flags)        # every sample exactly once
        if flags & pt.FLAG_REPRODUCIBLE:
            assert bit_equal(got, want), (sid, w, h, flags)                         # same schedule, same order: same bits
        else:
            # default mode: same samples, the order of the float additions into a pixel is the run's (include/pt_hip.h)
            S = max(1.0, float(np.abs(want).max()))
            assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2 * spp * 2.0 ** -24 * S, (sid, w, h, flags)


def test_static_entry_point_refuses_what_does_not_fit(renderer):
    torch = pytest.importorskip("torch")
    w, h = 1920, 1080                                                               # one partial image = 33 MB > 32 MiB
    renderer.upload(pt.Scene.builtin(1, w / h))
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    p = renderer.params(w, h, 0, 64, 5, seed=1, flags=pt.FLAG_REPRODUCIBLE)
    with pytest.raises(pt.PtError) as e:
        renderer.render_device_static(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert e.value.status == pt.PT_ERR_UNSUPPORTED and "static maximum" in str(e.value)
    # PT_FLAG_REFERENCE_QUERY would have to build (allocate) its tree on first use: refused until the caller has built it
    p = renderer.params(w, h, 0, 2, 5, seed=1, flags=pt.FLAG_REFERENCE_QUERY)
    with pytest.raises(pt.PtError) as e:
        renderer.render_device_static(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert e.value.status == pt.PT_ERR_UNSUPPORTED and "pt_hip_build_reference_tree" in str(e.value)
    # ... while the default mode needs no partial images and serves the same frame
    p = renderer.params(w, h, 0, 64, 5, seed=1, flags=0)
    renderer.render_device_static(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    img = acc.cpu().numpy()
    assert np.allclose(img[..., 3], 64, rtol=1e-6)
    # and a second, smaller frame right behind it on the same array
    w2, h2 = 320, 180
    renderer.upload(pt.Scene.builtin(1, w2 / h2))
    acc2 = torch.zeros((h2, w2, 4), dtype=torch.float32, device="cuda:0")
    acc3 = torch.zeros((h2, w2, 4), dtype=torch.float32, device="cuda:0")
    p2 = renderer.params(w2, h2, 0, 40, 50, seed=2, flags=pt.FLAG_REPRODUCIBLE)
    renderer.render_device_static(p2, acc2.data_ptr(), torch.cuda.current_stream().cuda_stream)
    renderer.render_device(p2, acc3.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bit_equal(acc2.cpu().numpy(), acc3.cpu().numpy())


def test_static_entry_point_leaves_device_memory_alone_and_records_into_a_graph(renderer):
    """No device allocation: the free-memory figure of the device does not move across static calls (first call at a new
    shape included).  And the call can be RECORDED: captured into a HIP graph (its timing events stay out of the graph),
    replayed twice, each replay leaves the frame pt_hip_render_device leaves."""
    torch = pytest.importorskip("torch")
    w, h, flags = 120, 68, pt.FLAG_REPRODUCIBLE
    renderer.upload(pt.Scene.builtin(1, w / h))
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    want = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    p = renderer.params(w, h, 0, 24, 50, seed=8, flags=flags)
    renderer.render_device(p, want.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for shape in ((w, h), (64, 36), (w, h)):
        q = renderer.params(shape[0], shape[1], 0, 24, 50, seed=8, flags=flags)
        renderer.render_device_static(q, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    assert bit_equal(acc.cpu().numpy(), want.cpu().numpy())
    # graph capture: the call made on the capturing stream; nothing runs until the replay
    acc.fill_(-3.0)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g.capture_begin()
        renderer.render_device_static(p, acc.data_ptr(), side.cuda_stream)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for _ in range(2):
        acc.fill_(-3.0)
        g.replay()
        torch.cuda.synchronize()
        assert bit_equal(acc.cpu().numpy(), want.cpu().numpy())
    del g
    # the context still serves ordinary calls, statistics included
    renderer.render_device(p, acc.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert renderer.stats()["samples"] == w * h * 24
