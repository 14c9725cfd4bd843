#!/usr/bin/env python3
"""BASELINE config 5 at its stated form: the five-fold ensemble (espnet_fold1-5 weights, each fold with its own mean/std
-> softmax -> mean over folds -> argmax; the definition is this build's) over EIGHT synthetic 40 000 x 40 000
slides, ONE SLIDE PER RANK (all five folds resident on every rank, no data-path collective; the per-slide
class totals are gathered once at the end).  `--gpus N` spawns the ranks like bench.py; rank r takes slides r, r+N, ...
Secondary measurement -- bench.py owns the headline.

    python tools/bench_ensemble.py [--gpus N] [--slides 8] [--size 40000]      ->  one JSON line (rank 0)

Per slide: the example-slide box pattern (tools/bench_slide.grid_boxes, 56 crops at 40k) is read from the synthetic
slide at the network size (1024 x 512, as SURVEY 8d prescribes for the synthetic crops), segmented by
gs_espnet_ensemble_forward in batches of 32, resized back to the crop size and max-composited on the slide's 1/8 map.
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

import bench_slide  # noqa: E402  (SynthSlide, grid_boxes, free_port: no torch import at module level)


def spawn(args):
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(bench_slide.free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = str(args.gpus)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL) for r in range(args.gpus)]
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    return 1 if any(rcs) else 0


def read_crop_at(slide, box, out_w, out_h):
    """the level-0 rectangle `box` sampled on an out_h x out_w grid (what an OpenSlide read + resize delivers), BGR"""
    import numpy as np
    x1, y1, x2, y2 = box
    # SynthSlide evaluates its field on any sampling grid; one downsample per axis
    sx, sy = (x2 - x1) / float(out_w), (y2 - y1) / float(out_h)
    xs = x1 + (np.arange(out_w, dtype=np.float64) + 0.5) * sx
    ys = y1 + (np.arange(out_h, dtype=np.float64) + 0.5) * sy
    return np.ascontiguousarray(slide.sample(xs, ys)[:, :, ::-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--slides", type=int, default=8)
    ap.add_argument("--size", type=int, default=40000)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(args)

    import numpy as np
    import torch
    from glomeruli_segmentation_amd.composite import SlideCompositor
    from glomeruli_segmentation_amd.engine import EspnetEngine, ensemble_segment, mask_resize_nearest
    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("GS_BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_reduce(t, op=None):
        kw = {} if op is None else {"op": op}
        if backend == "nccl" or not t.is_cuda:
            dist.all_reduce(t, **kw)
        else:
            tc = t.cpu()
            dist.all_reduce(tc, **kw)
            t.copy_(tc)

    S, NW, NH, B = args.size, 1024, 512, 32
    folds = [1, 2, 3, 4, 5]
    engines = []
    for f in folds:
        z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold%d.npz" % f))
        engines.append(EspnetEngine({k: z[k] for k in z.files}))
    mean_stds = [FOLD_MEAN_STD[f] for f in folds]
    example = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
    boxes = bench_slide.grid_boxes(S, example)
    mine = list(range(rank, args.slides, world))          # one slide per rank (and the next round of slides after it)

    totals = torch.zeros((args.slides, 5), dtype=torch.int64, device=dev)
    t_read = t_gpu = 0.0
    n_crops = 0
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for sid in mine:
        slide = bench_slide.SynthSlide(S, S, boxes, seed=sid)
        t0 = time.perf_counter()
        tiles = np.stack([read_crop_at(slide, b, NW, NH) for b in boxes])
        t_read += time.perf_counter() - t0
        t0 = time.perf_counter()
        comp = SlideCompositor(S, S, dev)
        for s in range(0, len(boxes), B):
            x = torch.from_numpy(tiles[s:s + B]).to(dev)
            mask, _ = ensemble_segment(engines, x, mean_stds)
            for j, b in enumerate(boxes[s:s + B]):
                m = mask_resize_nearest(mask[j], b[3] - b[1], b[2] - b[0])
                comp.paste(m, b[0], b[1])
                totals[sid] += torch.bincount(m.flatten().long(), minlength=5)[:5]
        torch.cuda.synchronize()
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
            "slides": args.slides, "crops_per_slide": len(boxes), "folds": len(folds),
            "gpu_leg_s": round(t_gpu_m, 3), "slides_per_s": round(args.slides / t_gpu_m, 2),
            "crops_per_s": round(crops_all / t_gpu_m, 1), "model_passes_per_s": round(crops_all * len(folds) / t_gpu_m, 1),
            "synthetic_region_generation_s": round(t_read_m, 3), "total_s": round(t_total_m, 3),
            "pixel_totals_per_slide": [[int(v) for v in row] for row in totals.tolist()],
            "note": "max over ranks; uploads, ensemble forward, mask resize, composite and counts are inside gpu_leg_s; "
                    "the region generator stands in for OpenSlide and is CPU numpy",
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
