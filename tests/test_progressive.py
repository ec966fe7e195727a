"""Prepare-once / commit-per-GPU uploads and progressive rendering through the C-ABI (VERDICT r2
next-round items 1b and 7).  GPU tests; the reference's only analogue of the latter is its progress print
(src/main.cpp:321-332), so the checks are internal identities: pieces add up to the frame, bit for bit in the
reproducible mode."""
import numpy as np
import pytest

import pathtracer_amd as pt
from tests._util import bit_equal

pytestmark = pytest.mark.gpu


def test_commit_of_a_prepared_scene_equals_upload(renderer):
    w, h = 96, 54
    for sid, flags in ((1, pt.FLAG_REPRODUCIBLE), (3, pt.FLAG_REPRODUCIBLE), (1, pt.FLAG_REPRODUCIBLE | pt.FLAG_FORCE_BVH), (1, pt.FLAG_REPRODUCIBLE | pt.FLAG_BRUTE_FORCE),
                       (2, pt.FLAG_REPRODUCIBLE | pt.FLAG_REFERENCE_QUERY)):
        scene = pt.Scene.builtin(sid, w / h)
        renderer.upload(scene)
        want = renderer.render(w, h, 0, 12, 50, seed=3, flags=flags)
        before = pt.tree_build_count()
        with scene.prepare() as prep:
            # one build, two commits (the second context stands in for a second GPU), the prepared scene freed before the renders
            with pt.Renderer(0) as other:
                renderer.commit(prep)
                other.commit(prep)
                assert pt.tree_build_count() == before + 1
                prep.close()
                got = renderer.render(w, h, 0, 12, 50, seed=3, flags=flags)
                got2 = other.render(w, h, 0, 12, 50, seed=3, flags=flags)
        assert bit_equal(got, want) and bit_equal(got2, want), (sid, flags)


def test_node_upload_builds_the_tree_once():
    scene = pt.Scene.builtin(2, 16 / 9)
    with pt.Node(1) as node:
        before = pt.tree_build_count()
        node.upload(scene)
        assert pt.tree_build_count() == before + 1


def test_node_reserve_and_the_pinned_image(renderer):
    """pt_node_reserve allocates a frame's buffers ahead of the render (the reference allocates its image before its timed
    span, main.cpp:385); the packed image device 0 writes into the node's pinned memory is the one pt_hip_resolve + pack
    give for the same sums -- at a first shape, at a bigger one (buffers re-grown) and at a smaller one (kept)."""
    scene = pt.Scene.builtin(1, 16 / 9)
    with pt.Node(1) as node:
        node.upload(scene)
        with pytest.raises(pt.PtError) as e:
            node.reserve(0, 4)
        assert e.value.status == pt.PT_ERR_INVALID
        node.reserve(64, 36)
        for w, h in ((64, 36), (160, 90), (32, 18)):
            sums, bgra = node.render(w, h, 0, 8, 10, seed=5, flags=pt.FLAG_REPRODUCIBLE, want_bgra=True)
            assert np.allclose(sums[..., 3], 8)
            want = pt.pack_bgra(pt.resolve(sums, 8)).reshape(h, w)
            assert (bgra == want).all(), (w, h)


@pytest.mark.parametrize("flags", [pt.FLAG_REPRODUCIBLE, 0, pt.FLAG_REPRODUCIBLE | pt.FLAG_FORCE_BVH])
def test_accumulate_flag_adds_pieces_into_one_buffer(renderer, flags):
    """pt_hip_render_device with PT_FLAG_ACCUMULATE: 4 pieces into one resident buffer == the pieces rendered apart and
    added in sequence (bit-equal with PT_FLAG_REPRODUCIBLE), and == the one-shot frame to summation order."""
    torch = pytest.importorskip("torch")
    w, h, depth, seed = 88, 56, 50, 9
    cuts = [0, 5, 6, 30, 41]                      # ragged pieces, one of a single sample
    renderer.upload(pt.Scene.builtin(1, w / h))
    stream = torch.cuda.current_stream().cuda_stream
    running = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    apart = np.zeros((h, w, 4), np.float32)
    for a, b in zip(cuts[:-1], cuts[1:]):
        renderer.render_device(renderer.params(w, h, a, b, depth, seed=seed, flags=flags | pt.FLAG_ACCUMULATE), running.data_ptr(), stream)
        torch.cuda.synchronize()
        piece = renderer.render(w, h, a, b, depth, seed=seed, flags=flags)
        apart = apart + piece                     # f32 adds, in the order of the pieces
        if flags & pt.FLAG_REPRODUCIBLE:
            assert bit_equal(running.cpu().numpy(), apart), (a, b)
    got = running.cpu().numpy()
    assert np.allclose(got[..., 3], cuts[-1], rtol=1e-6)   # every sample exactly once
    whole = renderer.render(w, h, 0, cuts[-1], depth, seed=seed, flags=flags)
    S = max(1.0, float(np.abs(whole).max()))
    assert np.abs(got.astype(np.float64) - whole.astype(np.float64)).max() <= 4 * cuts[-1] * 2.0 ** -24 * S


@pytest.mark.parametrize("force_collective", [False, True])
def test_node_progressive_pieces_and_snapshots(renderer, monkeypatch, force_collective):
    """pt_node_render(PT_FLAG_ACCUMULATE) x 4 + pt_node_snapshot: the running sums live on the GPU, a snapshot after every
    piece is the frame so far (its BGRA pixels are the resolve of its sums), rendering goes on after a snapshot, and a
    reset starts over.  With the collective forced the snapshot takes the out-of-place all-reduce path."""
    if force_collective:
        monkeypatch.setenv("PT_NODE_FORCE_COLLECTIVE", "1")
    w, h, depth, seed = 80, 45, 50, 4
    cuts = [0, 8, 9, 24, 40]
    scene = pt.Scene.builtin(1, w / h)
    renderer.upload(scene)
    apart = np.zeros((h, w, 4), np.float32)
    with pt.Node(1) as node:
        node.upload(scene)
        with pytest.raises(pt.PtError) as e:
            node.snapshot(w, h)
        assert e.value.status == pt.PT_ERR_NO_SCENE
        for a, b in zip(cuts[:-1], cuts[1:]):
            node.accumulate(w, h, a, b, depth, seed=seed, flags=pt.FLAG_REPRODUCIBLE)
            apart = apart + renderer.render(w, h, a, b, depth, seed=seed, flags=pt.FLAG_REPRODUCIBLE)
            sums, bgra, spp = node.snapshot(w, h)
            assert spp == b
            assert bit_equal(sums, apart), (a, b)
            assert (bgra.reshape(-1) == pt.pack_bgra(pt.resolve(apart, b))).all()
        # a piece of another shape is refused while sums are resident ...
        with pytest.raises(pt.PtError) as e:
            node.accumulate(w // 2, h, 40, 41, depth, seed=seed)
        assert e.value.status == pt.PT_ERR_INVALID
        # ... host outputs with the flag too; and after a reset the sums start from zero
        import ctypes as C
        p = pt.Params(w, h, 0, 1, depth, pt.FLAG_ACCUMULATE, seed)
        out = np.zeros((h, w, 4), np.float32)
        assert node.L.pt_node_render(node.h, C.byref(p), out.ctypes.data_as(C.c_void_p), None) == pt.PT_ERR_INVALID
        node.reset()
        node.accumulate(w // 2, h, 0, 3, depth, seed=seed, flags=pt.FLAG_REPRODUCIBLE)
        sums, _, spp = node.snapshot(w // 2, h, want_bgra=False)
        assert spp == 3
        renderer.upload(scene)
        assert bit_equal(sums, renderer.render(w // 2, h, 0, 3, depth, seed=seed, flags=pt.FLAG_REPRODUCIBLE))
        # a plain render forgets the resident sums
        node.render(w, h, 0, 2, depth, seed=seed)
        with pytest.raises(pt.PtError):
            node.snapshot(w, h)


@pytest.mark.parametrize("name,sid,n5,w,h,spp,depth", [
    ("C3_book1_1080p_4096spp", 1, 0, 1920, 1080, 4096, 50),
    ("C4_motion_blur_4k_1024spp", 3, 0, 3840, 2160, 1024, 50),
    ("C5_1e6_spheres_4k_1024spp", 5, 1000000, 3840, 2160, 1024, 50),
])
def test_full_size_configs_by_properties(renderer, name, sid, n5, w, h, spp, depth):
    """BASELINE.json's configurations at their FULL sizes (8.5 G samples each; the oracle cannot follow there), through
    properties that do not depend on size: every sample exactly once (alpha of every returned colour is 1, and sums of
    ones are exact in f32 up to 2^24: the alpha plane must equal spp everywhere), the frame in one piece == the 8 per-GPU
    slices of SURVEY.md 8(e) accumulated into one buffer (same sample set, float summation order apart), no NaN / negative
    sums, and the statistics the roofline uses are those of the sample count."""
    torch = pytest.importorskip("torch")
    renderer.upload(pt.Scene.builtin(sid, w / h, n5))
    stream = torch.cuda.current_stream().cuda_stream
    whole = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    renderer.render_device(renderer.params(w, h, 0, spp, depth, seed=2), whole.data_ptr(), stream)
    torch.cuda.synchronize()
    st = renderer.stats()
    assert st["samples"] == w * h * spp and st["segments"] >= st["samples"]
    assert bool((whole[..., 3] == float(spp)).all()), name
    assert bool(torch.isfinite(whole).all()) and float(whole.min()) >= 0.0
    pieces = torch.zeros_like(whole)
    segs = 0
    for g in range(8):
        a, b = pt.spp_slice(0, spp, 8, g)
        renderer.render_device(renderer.params(w, h, a, b, depth, seed=2, flags=pt.FLAG_ACCUMULATE), pieces.data_ptr(), stream)
        torch.cuda.synchronize()
        segs += renderer.stats()["segments"]
    assert segs == st["segments"], name                        # the same paths, segment for segment
    assert bool((pieces[..., 3] == float(spp)).all()), name
    rel = float(((pieces[..., :3] - whole[..., :3]).abs() / whole[..., :3].abs().clamp_min(1.0)).max())
    assert rel <= 64 * 2.0 ** -24 * 8, (name, rel)             # summation order only
    print(f"{name}: {st['samples'] / 1e9:.2f} G samples in {st['kernel_ms']:.0f} ms = {st['samples'] / st['kernel_ms'] / 1e3:.0f} Msamples/s; "
          f"one piece vs 8 accumulated slices: max rel diff {rel:.2e}")
