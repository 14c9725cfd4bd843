"""Multi-GPU sharding of the per-patch path: one process per GPU, a contiguous tile range per
rank, and ONE exchange at the end of a slide (masks gathered to rank 0, per-class pixel counts
summed) -- torch.distributed with the "nccl" backend is RCCL over xGMI on ROCm; "gloo" runs the
same code on CPU for the tests.

The reference has no distributed code at all: its loop
(VisualizeResults_iou.py:100) has no cross-iteration state except CSV appends and one confusion
matrix, which is why a tile range per rank with no data-path collective is exact.
"""
import numpy as np


def rank_range(total, rank, world):
    """Contiguous range [lo, hi) of `total` row-major tiles owned by `rank` of `world`:
    rank r takes [r*T/R, (r+1)*T/R)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    return (total * rank) // world, (total * (rank + 1)) // world


def plan_grid(height, width, tile_h, tile_w):
    """Row-major sliding-window plan over a canvas: list of (y, x) origins; the last row/column is
    shifted inwards so every window is full-size."""
    ys = list(range(0, max(height - tile_h, 0) + 1, tile_h))
    xs = list(range(0, max(width - tile_w, 0) + 1, tile_w))
    if ys[-1] + tile_h < height:
        ys.append(height - tile_h)
    if xs[-1] + tile_w < width:
        xs.append(width - tile_w)
    return [(y, x) for y in ys for x in xs]


def segment_sharded(compute, load_tiles, total, rank, world, dist=None, device=None, batch=32, classes=5,
                    gather_masks=True):
    """Run `compute` over this rank's tile range and do the final exchange.

    compute(tiles uint8 [n,H,W,3]) -> (masks uint8 [n,H,W], counts int64 [n,classes])  (numpy or torch)
    load_tiles(lo, hi) -> uint8 [hi-lo,H,W,3] for global tile indices [lo, hi)
    Returns (masks [total,H,W] on rank 0 else None, counts_total int64 [classes] on every rank).
    """
    import torch
    lo, hi = rank_range(total, rank, world)
    masks, counts = [], np.zeros(classes, dtype=np.int64)
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        m, c = compute(load_tiles(s, e))
        m = m.cpu().numpy() if hasattr(m, "cpu") else np.asarray(m)
        c = c.cpu().numpy() if hasattr(c, "cpu") else np.asarray(c)
        masks.append(m)
        counts += c.reshape(-1, classes).sum(0).astype(np.int64)
    local = np.concatenate(masks, 0) if masks else None
    if world == 1 or dist is None:
        return local, counts
    dev = device if device is not None else torch.device("cpu")
    tot = torch.from_numpy(counts).to(dev)
    dist.all_reduce(tot)                                   # per-class pixel totals of the whole slide
    out = None
    if gather_masks:
        # ranges differ by at most one tile: pad to the longest so one gather moves everything
        shape = torch.tensor(list(local.shape[1:]) if local is not None else [0, 0], dtype=torch.int64, device=dev)
        dist.all_reduce(shape, op=dist.ReduceOp.MAX)
        th, tw = int(shape[0]), int(shape[1])
        longest = max(rank_range(total, r, world)[1] - rank_range(total, r, world)[0] for r in range(world))
        buf = torch.zeros((longest, th, tw), dtype=torch.uint8, device=dev)
        if local is not None and len(local):
            buf[:hi - lo] = torch.from_numpy(local).to(dev)
        recv = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, recv, dst=0)
        if rank == 0:
            parts = []
            for r in range(world):
                rlo, rhi = rank_range(total, r, world)
                parts.append(recv[r][:rhi - rlo].cpu().numpy())
            out = np.concatenate(parts, 0)
    return out, tot.cpu().numpy()
