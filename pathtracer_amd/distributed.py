"""Multi-GPU path: samples-per-pixel split across ranks, one all-reduce (sum) of the float accumulation buffer, resolve on
rank 0.  One process per GPU; torch.distributed is only the transport (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" on CPU for the tests).  The reference has no counterpart: its only parallelism is image tiles over threads
(src/main.cpp:415-469); what makes the split exact here is the RNG keyed by (pixel, GLOBAL sample index), so the set of
samples does not depend on how many ranks rendered it.

There is no data-path collective besides that one all-reduce: 1920x1080x4 floats = 33 MB per frame.
"""
from typing import Callable, Tuple


def spp_slice(rank: int, world: int, spp_total: int) -> Tuple[int, int]:
    """Contiguous global sample indices [begin, end) of `rank`; the first spp_total % world ranks take one extra."""
    if world < 1 or not (0 <= rank < world) or spp_total < 0:
        raise ValueError("bad rank/world/spp")
    q, r = divmod(spp_total, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def render_frame(render_slice: Callable[[int, int], "torch.Tensor"], spp_total: int, resolve: Callable[["torch.Tensor", int], object] = None,
                 group=None):
    """render_slice(begin, end) -> (H, W, 4) float32 tensor holding this rank's per-pixel SUM over its samples
    (on the device the process group's backend reduces: CUDA for nccl/RCCL, CPU for gloo).
    Returns resolve(total_sum, spp_total) on rank 0 (the total sum itself if resolve is None) and None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, end = spp_slice(rank, world, spp_total)
    if end > begin:
        acc = render_slice(begin, end)
    else:  # more ranks than samples: contribute zeros of the right shape
        acc = torch.zeros_like(render_slice(0, 1))
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    if rank != 0:
        return None
    return resolve(acc, spp_total) if resolve else acc


def hip_slice_renderer(renderer, width: int, height: int, max_depth: int, seed: int = 0, flags: int = 0, device=None):
    """render_slice for render_frame backed by libpt_hip.so: renders into a torch CUDA tensor on torch's current stream."""
    import torch
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    acc = torch.empty((height, width, 4), dtype=torch.float32, device=dev)

    def render_slice(begin: int, end: int):
        p = renderer.params(width, height, begin, end, max_depth, seed, flags)
        renderer.render_device(p, acc.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return acc

    return render_slice


def render_progressive(render_slice: Callable[[int, int], "torch.Tensor"], spp_total: int, passes: int,
                       on_snapshot: Callable[["torch.Tensor", int, int], None] = None, spp_done: int = 0, total: "torch.Tensor" = None,
                       group=None):
    """Progressive / resumable accumulation.  The sample range [spp_done, spp_total) is rendered in
    `passes` contiguous pieces; each piece is split across the ranks like a frame (spp_slice) and all-reduced, then added
    to the running per-pixel sum.  After every piece rank 0 calls on_snapshot(running_sum, samples_so_far, pass_index) --
    resolve / write a preview there.  Because the RNG is keyed by the GLOBAL sample index, the union of the pieces is
    exactly the sample set of a one-shot frame, and a render can be resumed later from (running sum, samples so far):
    pass them back as `total` and `spp_done`.  Returns (running_sum, spp_total) on rank 0 and (None, spp_total) elsewhere;
    the reference's only analogue is its progress printout (src/main.cpp:321-332)."""
    import torch
    import torch.distributed as dist
    if passes < 1 or spp_total < spp_done:
        raise ValueError("bad passes / sample range")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    todo = spp_total - spp_done
    done = spp_done
    for k in range(passes):
        lo = spp_done + todo * k // passes
        hi = spp_done + todo * (k + 1) // passes
        if hi == lo:
            continue
        b, e = spp_slice(rank, world, hi - lo)
        if e > b:
            acc = render_slice(lo + b, lo + e)
        else:
            acc = torch.zeros_like(render_slice(lo, lo + 1))
        if world > 1:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        done = hi
        if rank == 0:
            total = acc.clone() if total is None else total.add_(acc)
            if on_snapshot:
                on_snapshot(total, done, k)
    return (total if rank == 0 else None), done
