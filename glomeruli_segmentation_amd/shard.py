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


def init_from_env(use_gpu=True):
    """Process-group setup of a CLI started one process per GPU (torchrun, or launch.spawn_ranks): RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the environment.  Returns (rank, world, local_rank, dist | None); with one rank nothing is
    initialised.  Backend "nccl" (RCCL over xGMI) on the GPUs, "gloo" without them; GLOMSEG_DIST_BACKEND overrides (the
    one-GPU rehearsal of the tests uses gloo with every rank on device 0)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world <= 1:
        return 0, 1, local, None
    from .launch import place_rank
    place_rank()                                   # before the first GPU call
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("GLOMSEG_DIST_BACKEND") or ("nccl" if use_gpu and torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if use_gpu and torch.cuda.is_available() and os.environ.get("GLOMSEG_ONE_GPU") != "1":
        log_device_order(local)
    return rank, world, local, dist


def log_device_order(local):
    """after the device exists: is HIP device `local` the GPU place_rank pinned this rank's CPUs for?  (stderr; never fatal)"""
    import sys
    import torch
    from .launch import check_device_order
    try:
        pr = torch.cuda.get_device_properties(local)
        ok, msg = check_device_order(local, pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception as e:      # an attribute this torch lacks, a device that went away: placement is a speed matter only
        ok, msg = None, "device order not checked (%s)" % e
    if ok is False:
        print("WARNING rank placement: " + msg, file=sys.stderr)
    return ok, msg


def all_reduce_any(t, dist, op=None):
    """dist.all_reduce of a tensor wherever it lives: as it is under "nccl" (RCCL takes device tensors, GPU to GPU over xGMI),
    staged through the host under "gloo" (the CPU tests and the one-GPU rehearsal), which cannot take them.  In place."""
    if dist is None:
        return t
    kw = {} if op is None else {"op": op}
    if dist.get_backend() != "gloo" or not t.is_cuda:
        dist.all_reduce(t, **kw)
    else:
        tc = t.cpu()
        dist.all_reduce(tc, **kw)
        t.copy_(tc)
    return t


def finish_ranks(dist):
    """Success path of a sharded CLI: every rank has made the same collectives, so a barrier and the teardown are safe."""
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def abort_rank(dist):
    """Failure path (call from an `except` block): this rank failed ALONE -- an unreadable image, a shape assert, out of memory --
    while its peers sit in a gather / all-reduce.  A barrier here would be a collective the peers never reach: this rank would
    block until the backend's timeout instead of exiting, and launch.spawn_ranks, which ends the job when a child exits
    non-zero, would never see it.  So: print the traceback, skip barrier and teardown (callers close their GPU handles only AFTER
    this returns, i.e. in single-process runs: a device synchronise after a GPU fault may never come back), leave with status 1
    (a SystemExit's own non-zero code is kept) -- at once, without
    interpreter shutdown (a process-group destructor may itself wait for the peers).  With one rank the exception just
    propagates.

    EVERY exit from inside the sharded region is a failure of the job, a `SystemExit(0)`, `SystemExit(None)` or
    `SystemExit("message")` included: the rank's peers are in (or on their way to) a collective that this rank will now
    never join, so there is no "clean early finish" of one rank -- a rank that has nothing to do takes part in the exchange
    with an empty range instead (rank_range).  Those exits therefore leave with status 1 as well (and their traceback),
    which is what makes launch.spawn_ranks end the peers; only a non-zero integer code is passed through unchanged."""
    if dist is None:
        return
    import os
    import sys
    import traceback
    exc = sys.exc_info()[1]
    code = exc.code if isinstance(exc, SystemExit) and isinstance(exc.code, int) and exc.code != 0 else 1   # a SystemExit keeps its own code
    traceback.print_exc()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)


def collective_device(dist, device=None):
    """the device tensors handed to `dist` must live on: the CPU for gloo, this rank's current GPU otherwise"""
    import torch
    if dist is None or dist.get_backend() == "gloo":
        return torch.device("cpu")
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def gather_rows(rows, rank, world, dist, dst=0):
    """Per-rank lists (CSV rows, detection boxes: a few KB) -> their concatenation in rank order on `dst`, None elsewhere.
    Ranks own contiguous ranges, so rank order is the single-process order."""
    if dist is None or world == 1:
        return list(rows)
    parts = [None] * world if rank == dst else None
    dist.gather_object(list(rows), parts, dst=dst)
    return [r for part in parts for r in part] if rank == dst else None


def reduce_sum_to_all(array, dist, device=None):
    """element-wise sum over ranks of a small integer / float array (confusion matrix, per-class totals)"""
    import torch
    if dist is None:
        return np.asarray(array)
    t = torch.from_numpy(np.ascontiguousarray(array)).to(collective_device(dist, device))
    dist.all_reduce(t)
    return t.cpu().numpy()


def segment_sharded(compute, load_tiles, total, rank, world, dist=None, device=None, batch=32, classes=None,
                    gather_masks=True):
    """Run `compute` over this rank's tile range and do the final exchange.

    compute(tiles uint8 [n,H,W,3]) -> (masks uint8 [n,H,W], counts int64 [n,classes])  (numpy or torch)
    load_tiles(lo, hi) -> uint8 [hi-lo,H,W,3] for global tile indices [lo, hi)
    Returns (masks [total,H,W] on rank 0 else None, counts_total int64 [classes] on every rank).
    `classes` = the model's class count (Model.py:311; engine.classes); None takes it from the counts `compute` returns -- a rank
    whose range is empty then learns it from the other ranks (a MAX all-reduce) before the totals are summed.

    Masks that `compute` returns as device tensors stay on the device: they are concatenated there and handed to
    the collective as they are (backend "nccl" = RCCL: GPU to GPU over xGMI); only the gathered result on rank 0
    is copied to the host.  With the "gloo" backend (CPU tests, one-GPU rehearsal) device tensors are staged
    through host memory because that backend cannot take them.
    """
    import torch
    lo, hi = rank_range(total, rank, world)
    masks = []
    counts = None
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        m, c = compute(load_tiles(s, e))
        m = m if isinstance(m, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(m))
        c = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c))
        masks.append(m)
        if classes is None:
            if c.dim() != 2:
                raise ValueError("segment_sharded: compute must return counts [n, classes] when `classes` is not given")
            classes = int(c.shape[1])
        csum = c.reshape(-1, classes).sum(0).to(torch.int64)
        counts = csum if counts is None else counts + csum
    local = torch.cat(masks, 0) if masks else None
    if dist is None:      # (a process group of one rank still runs the exchange: the nccl test executes it on one GPU)
        if counts is None:
            counts = torch.zeros(classes or 0, dtype=torch.int64)
        return (local.cpu().numpy() if local is not None else None), counts.cpu().numpy()
    # gloo takes host tensors; every other backend (nccl = RCCL) takes tensors on this rank's GPU -- also from a rank whose
    # range is empty or whose compute returned host arrays
    dev = collective_device(dist, device)
    ncls = torch.tensor([classes or 0], dtype=torch.int64, device=dev)
    dist.all_reduce(ncls, op=dist.ReduceOp.MAX)          # ranks with an empty range take the class count from the others
    if classes is not None and int(ncls[0]) != classes:
        raise ValueError("segment_sharded: ranks disagree about the class count (%d here, %d elsewhere)" % (classes, int(ncls[0])))
    classes = int(ncls[0])
    if counts is None:
        counts = torch.zeros(classes, dtype=torch.int64)
    tot = counts.to(dev)
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
            buf[:hi - lo] = local.to(dev)                  # device-to-device when the masks already live there
        recv = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, recv, dst=0)
        if rank == 0:
            parts = []
            for r in range(world):
                rlo, rhi = rank_range(total, r, world)
                parts.append(recv[r][:rhi - rlo])
            out = torch.cat(parts, 0).cpu().numpy()        # the one device-to-host copy, on rank 0
    return out, tot.cpu().numpy()
