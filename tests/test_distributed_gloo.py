"""CPU, world_size 2 and 3 over gloo: the spp split + all-reduce + resolve logic of pathtracer_amd/distributed.py.
The per-rank renderer is injected; here it is the oracle's Philox render (tests may use the oracle as a stand-in for the
GPU, the product may not), so the test checks exactly what the multi-GPU bench relies on: slices partition the samples,
the reduced sum equals a single-rank render of all samples, only rank 0 resolves."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd.distributed import spp_slice, render_frame, render_progressive

W, H, SPP, DEPTH, SEED = 40, 24, 7, 8, 11


def test_spp_slices_partition():
    for world in (1, 2, 3, 4, 8):
        for spp in (0, 1, 5, 8, 512, 4096, 4099):
            parts = [spp_slice(r, world, spp) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - b for b, e in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        spp_slice(2, 2, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    import pathtracer_amd as pt
    from tests._util import to_scene_arrays
    O = po.Oracle()
    h = O.scene(to_scene_arrays(pt.Scene.builtin(3, W / H)))   # motion-blur scene: time-sampled rays too

    def render_slice(b, e):
        img, _ = O.render_philox(h, W, H, b, e, DEPTH, seed=SEED, nthreads=2)
        return torch.from_numpy(img)

    res = render_frame(render_slice, SPP, resolve=lambda acc, spp: torch.from_numpy(pt.resolve(acc.numpy(), spp)).reshape(H, W, 4))
    assert (res is None) == (rank != 0)
    if rank == 0:
        np.save(out_path, res.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_reduce_resolve_equals_single_rank(world, tmp_path, oracle):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    import pathtracer_amd as pt
    from tests._util import to_scene_arrays
    h = oracle.scene(to_scene_arrays(pt.Scene.builtin(3, W / H)))
    full, _ = oracle.render_philox(h, W, H, 0, SPP, DEPTH, seed=SEED)
    oracle.scene_free(h)
    want = pt.resolve(full, SPP).reshape(H, W, 4)
    # same samples, summed in a different association (per-rank partial sums): float noise only
    assert np.abs(got - want).max() <= 4e-6
    assert got.min() >= 0 and got.max() <= 1


def _progressive_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    import pathtracer_amd as pt
    from tests._util import to_scene_arrays
    O = po.Oracle()
    h = O.scene(to_scene_arrays(pt.Scene.builtin(3, W / H)))
    calls = []

    def render_slice(b, e):
        calls.append((b, e))
        img, _ = O.render_philox(h, W, H, b, e, DEPTH, seed=SEED, nthreads=2)
        return torch.from_numpy(img)

    snaps = []
    # 11 samples: first a 2-pass run over [0, 5), then resumed from its state for [5, 11) in 3 passes
    total, done = render_progressive(render_slice, 5, 2, on_snapshot=lambda t, n, k: snaps.append((n, float(t[..., 3].mean()))))
    total, done = render_progressive(render_slice, 11, 3, on_snapshot=lambda t, n, k: snaps.append((n, float(t[..., 3].mean()))), spp_done=done, total=total)
    assert done == 11 and (total is None) == (rank != 0)
    if rank == 0:
        assert [n for n, _ in snaps] == [2, 5, 7, 9, 11]
        assert all(abs(a - n) < 1e-4 for n, a in snaps)          # alpha sums to the samples so far: every sample exactly once
        np.save(out_path, total.numpy())
    else:
        assert snaps == []
    dist.barrier()
    dist.destroy_process_group()


def test_progressive_resumable_accumulation_equals_one_shot(tmp_path, oracle):
    out = str(tmp_path / "sum.npy")
    mp.spawn(_progressive_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    import pathtracer_amd as pt
    from tests._util import to_scene_arrays
    h = oracle.scene(to_scene_arrays(pt.Scene.builtin(3, W / H)))
    full, _ = oracle.render_philox(h, W, H, 0, 11, DEPTH, seed=SEED)
    oracle.scene_free(h)
    assert np.abs(got - full).max() <= 2 * 11 * 2.0 ** -24 * max(1.0, float(np.abs(full).max()))
