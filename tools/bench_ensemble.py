#!/usr/bin/env python3
"""BASELINE config 5 at its stated form: the five-fold ensemble (espnet_fold1-5 weights, each fold with its own mean/std
-> softmax -> mean over folds -> argmax; the definition is this build's) over EIGHT synthetic 40 000 x 40 000
slides, ONE SLIDE PER RANK (all five folds resident on every rank, no data-path collective; the per-slide
class totals are gathered once at the end).  `--gpus N` spawns the ranks like bench.py; rank r takes slides r, r+N, ...
Secondary measurement -- bench.py owns the headline.

    python tools/bench_ensemble.py [--gpus N] [--slides 8] [--size 40000]      ->  one JSON line (rank 0)

Per slide: the example-slide box pattern (tools/bench_slide.grid_boxes, 56 crops at 40k) is read from the synthetic
slide at level 0 -- every crop at its own size, as make_seg_data.py:357-361 cuts them -- and goes through ONE call of the
batched crop pipeline with the five members (gs_espnet_segment_crops_host: every member resamples the crops with its own
mean/std, adds its probabilities in the decoder tail; the last member's tail writes the masks; resize back, counts and the
max-composite on the slide's 1/8 map are batched launches of the same call).
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import bench_slide  # noqa: E402  (SynthSlide, grid_boxes: no torch import at module level)


def spawn(args):
    from glomeruli_segmentation_amd.launch import spawn_ranks
    return spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--slides", type=int, default=8)
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only with CPU stand-ins over gloo (tools/dry.py): the CPU test suite's 8-rank rehearsal; never a measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(args)

    import numpy as np
    import torch
    from glomeruli_segmentation_amd.launch import place_rank
    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD

    place_rank()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("GS_BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    backend = "gloo" if dry else os.environ.get("GS_BENCH_BACKEND", "nccl")
    if dry:
        import dry as stand_ins          # tools/dry.py (this directory is on sys.path: bench_slide is imported from it)
        SlideCompositor, segment_crops_host = stand_ins.DryCompositor, stand_ins.dry_segment_crops_host
        dev = torch.device("cpu")
    else:
        from glomeruli_segmentation_amd.composite import SlideCompositor
        from glomeruli_segmentation_amd.engine import EspnetEngine, segment_crops_host
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

    def sync():
        if not dry:
            torch.cuda.synchronize()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from glomeruli_segmentation_amd.shard import all_reduce_any, log_device_order
    if world > 1 and os.environ.get("GS_BENCH_ONE_GPU") != "1" and not getattr(args, "dry_run", False):
        log_device_order(local)      # is HIP device `local` the GPU place_rank pinned this rank's CPUs for? (stderr, never fatal)

    def all_reduce(t, op=None):      # one helper owns the backend choice (device tensors as they are under RCCL, via the host under gloo)
        all_reduce_any(t, dist, op)

    S, NW, NH, B = args.size, 1024, 512, 32
    folds = [1, 2, 3, 4, 5]
    engines = []
    for f in folds:
        if dry:
            engines.append(stand_ins.DryEngine())
            continue
        z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold%d.npz" % f))
        engines.append(EspnetEngine({k: z[k] for k in z.files}, lanes=2))
    mean_stds = [FOLD_MEAN_STD[f] for f in folds]
    example = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
    boxes = bench_slide.grid_boxes(S, example)
    mine = list(range(rank, args.slides, world))          # one slide per rank (and the next round of slides after it)

    totals = torch.zeros((args.slides, int(engines[0].classes)), dtype=torch.int64, device=dev)
    t_read = t_gpu = 0.0
    n_crops = 0
    origins = [(b[0], b[1]) for b in boxes]
    sync()
    t_start = time.perf_counter()
    warm = False
    for sid in mine:
        slide = bench_slide.SynthSlide(S, S, boxes, seed=sid)
        t0 = time.perf_counter()
        crops = [np.ascontiguousarray(slide.read_region(b[0], b[1], b[2] - b[0], b[3] - b[1], 1.0)[:, :, ::-1]) for b in boxes]
        t_read += time.perf_counter() - t0
        if not warm:      # workspaces and pinned staging are allocated once per process, outside the timed leg
            segment_crops_host(engines, mean_stds, crops, NH, NW, B, want_masks=True)      # (both lanes of every member)
            warm = True
        t0 = time.perf_counter()
        comp = SlideCompositor(S, S, dev)
        r = segment_crops_host(engines, mean_stds, crops, NH, NW, B, want_masks=True, paste=comp.paste_target(), origins=origins)
        totals[sid] += torch.from_numpy(r["counts"].sum(0)).to(dev)
        sync()
        t_gpu += time.perf_counter() - t0
        n_crops += len(boxes)
    t_total = time.perf_counter() - t_start
    if dist is not None:
        all_reduce(totals)                                 # the one exchange: per-slide class totals (8 x 5 integers)

    def mx(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    t_gpu_m, t_read_m, t_total_m = mx(t_gpu), mx(t_read), mx(t_total)
    if rank == 0:
        crops_all = len(boxes) * args.slides
        print(json.dumps({
            "config": "cfg 5: five-fold ensemble over %d synthetic %d x %d slides, one slide per rank, %d rank(s)" % (args.slides, S, S, world),
            "data": "dry-run (CPU stand-ins, no device work)" if dry else "synthetic",
            "slides": args.slides, "crops_per_slide": len(boxes), "folds": len(folds),
            "gpu_leg_s": round(t_gpu_m, 3), "slides_per_s": round(args.slides / t_gpu_m, 2),
            "crops_per_s": round(crops_all / t_gpu_m, 1), "model_passes_per_s": round(crops_all * len(folds) / t_gpu_m, 1),
            "synthetic_region_generation_s": round(t_read_m, 3), "total_s": round(t_total_m, 3),
            "pixel_totals_per_slide": [[int(v) for v in row] for row in totals.tolist()],
            "note": "max over ranks; pageable level-0 crops in, crop-size maps out: staging, uploads, five resample + forward passes "
                    "per batch, mask resize, composite, counts and downloads are inside gpu_leg_s; the region generator stands in "
                    "for OpenSlide and is CPU numpy",
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
