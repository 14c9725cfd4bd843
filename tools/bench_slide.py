#!/usr/bin/env python3
"""BASELINE config 4 at its stated size: detect -> merge -> crop -> segment -> composite over ONE synthetic
40 000 x 40 000 whole-slide image (mpp 0.2277, objective 40x, detector windows of 2000 um with 0.1 overlap read
at downsample 8, "detections" = the 28 annotated boxes of the reference's example slide repeated on a grid), tile-sharded
over the ranks (one process per GPU, `--gpus N` spawns them like bench.py; no data-path collective, one all-reduce of
the 1/8 map and the per-class totals at the end).  Secondary measurement -- bench.py owns the headline.

    python tools/bench_slide.py [--gpus N] [--size 40000]       ->  one JSON line (rank 0)

The canvas is never materialised: `SynthSlide.read_region` evaluates a seeded field (stain-coloured base + Gaussian
blobs at the box centres + hashed pixel noise) on the sampling grid that is asked for, like OpenSlide would decode a region.
The detector runs for real (gs_detector_forward, synthetic weights -- the reference's graph is external) on every
window, and its boxes go through the reference's thresholding / CSV arithmetic; since untrained weights put boxes
anywhere, the boxes that are cropped and segmented are the example-slide pattern, as SURVEY 8d prescribes.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class SynthSlide:
    def __init__(self, width, height, boxes, seed=0):
        import numpy as np
        self.w, self.h = width, height
        rng = np.random.default_rng(seed)
        b = np.asarray(boxes, dtype=np.float64)
        self.cx, self.cy = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
        self.sig = np.maximum(b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]) / 4.0
        self.col = rng.standard_normal((len(b), 3)) * np.array([28.0, 43.0, 20.0]) * 1.5      # RGB offsets

    def read_region(self, x, y, w, h, ds):
        """uint8 RGB [h,w,3] of the level-0 rectangle starting at (x,y), sampled every `ds` pixels"""
        import numpy as np
        return self.sample(x + np.arange(w, dtype=np.float64) * ds, y + np.arange(h, dtype=np.float64) * ds)

    def sample(self, xs, ys):
        """uint8 RGB [len(ys),len(xs),3] of the field at the level-0 coordinates xs (columns) x ys (rows), both ascending"""
        import numpy as np
        h, w = len(ys), len(xs)
        img = np.empty((h, w, 3), dtype=np.float32)
        img[:] = np.array([199.0, 170.0, 204.0], dtype=np.float32)
        x, y, x1, y1 = xs[0], ys[0], xs[-1], ys[-1]
        near = (self.cx + 4 * self.sig > x) & (self.cx - 4 * self.sig < x1) & (self.cy + 4 * self.sig > y) & (self.cy - 4 * self.sig < y1)
        for k in np.nonzero(near)[0]:
            gx = np.exp(-((xs - self.cx[k]) ** 2) / (2 * self.sig[k] ** 2)).astype(np.float32)
            gy = np.exp(-((ys - self.cy[k]) ** 2) / (2 * self.sig[k] ** 2)).astype(np.float32)
            img += (gy[:, None] * gx[None, :])[:, :, None] * self.col[k].astype(np.float32)
        # hashed noise: a function of the level-0 coordinate only, so overlapping reads agree
        hx = (xs.astype(np.int64) * 73856093) & 0xFFFF
        hy = (ys.astype(np.int64) * 19349663) & 0xFFFF
        n = ((hy[:, None] ^ hx[None, :]) * 2654435761 >> 7) & 0x1F
        img += (n.astype(np.float32) - 15.5)[:, :, None] * 0.6
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def grid_boxes(size, example):
    """the example slide's 28 annotated boxes (a 53 248 x 23 040 level-0 field): positions scaled to the canvas width,
    sizes kept, the band repeated down the canvas"""
    import numpy as np
    ex = np.asarray(example, dtype=np.int64)
    fw, fh = 53248, 23040
    sc = size / float(fw)
    band = int(fh * sc)
    out = []
    for oy in range(0, size, band):
        for x1, y1, x2, y2 in ex:
            cx, cy = (x1 + x2) / 2 * sc, (y1 + y2) / 2 * sc + oy
            w, h = x2 - x1, y2 - y1
            bx1, by1 = int(cx - w / 2) // 8 * 8, int(cy - h / 2) // 8 * 8
            if bx1 >= 0 and by1 >= 0 and bx1 + w < size and by1 + h < size:
                out.append([bx1, by1, bx1 + int(w), by1 + int(h)])
    return out


def spawn(args):
    from glomeruli_segmentation_amd.launch import spawn_ranks
    return spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--detector-batch", type=int, default=12, help="windows per detector forward (36 windows = 3 x 12: no ragged last batch)")
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only (spawn, rank ranges, reductions, JSON) with CPU stand-ins for the engine, the detector and "
                         "the compositor over gloo (tools/dry.py): what the CPU test suite runs with 8 ranks; never a measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(args)

    import numpy as np
    import torch
    from glomeruli_segmentation_amd import detect, merge
    from glomeruli_segmentation_amd.pipeline import segment_crops
    from glomeruli_segmentation_amd.shard import rank_range
    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD

    from glomeruli_segmentation_amd.launch import place_rank
    place_rank()           # CPU share of this rank's GPU, before the first GPU call
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("GS_BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    backend = "gloo" if dry else os.environ.get("GS_BENCH_BACKEND", "nccl")
    if dry:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import dry as stand_ins
        SlideCompositor = stand_ins.DryCompositor
        dev = torch.device("cpu")
    else:
        from glomeruli_segmentation_amd.composite import SlideCompositor
        from glomeruli_segmentation_amd.detector import FrcnnDetector, synthetic_weights
        from glomeruli_segmentation_amd.engine import EspnetEngine
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    from glomeruli_segmentation_amd.shard import all_reduce_any, log_device_order
    if world > 1 and os.environ.get("GS_BENCH_ONE_GPU") != "1" and not dry:
        log_device_order(local)

    def all_reduce(t, op=None):
        all_reduce_any(t, dist, op)

    S, mpp = args.size, 0.2277
    example = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
    boxes_all = grid_boxes(S, example)
    slide = SynthSlide(S, S, boxes_all)
    mean, std = FOLD_MEAN_STD[1]
    if dry:
        eng, det = stand_ins.DryEngine(), stand_ins.DryDetector()
    else:
        z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
        eng = EspnetEngine({k: z[k] for k in z.files}, lanes=2)
        det = FrcnnDetector(synthetic_weights(0))
    t_start = time.perf_counter()

    # ---- the slide's regions, generated once (the generator stands in for OpenSlide and is CPU numpy)
    level, ds = detect.pick_level(40, (1.0, 2.0, 4.0, 8.0))
    plan = detect.plan_windows(S, S, mpp, mpp, ds, 2000, 0.1)
    wins = plan.origins()
    lo, hi = rank_range(len(wins), rank, world)
    t0 = time.perf_counter()
    regions = [slide.read_region(xs, ys, plan.window_x, plan.window_y, ds) for (_, _, xs, ys) in wins[lo:hi]]
    blo, bhi = rank_range(len(boxes_all), rank, world)
    mine = boxes_all[blo:bhi]
    crops = [np.ascontiguousarray(slide.read_region(b[0], b[1], b[2] - b[0], b[3] - b[1], 1.0)[:, :, ::-1]) for b in mine]   # make_seg_data.py:357-361
    t_read = time.perf_counter() - t0

    def detect_leg():
        # this rank's windows (detect_glomus_test.py:270-284) through the detector's host pipeline, sixteen per forward
        it = iter(regions)
        return detect.scan_slide(lambda xs, ys, w, h: next(it), det, plan, 0.2, "site", "slide", "slide.ndpi", rank=rank, world=world,
                                 batch=args.detector_batch)

    def segment_leg():
        # crop -> resample -> segment -> resize back -> count -> composite: one pipeline call (gs_espnet_segment_crops_host)
        comp = SlideCompositor(S, S, dev)
        counts = torch.zeros(int(eng.classes), dtype=torch.int64, device=dev)
        if crops:
            _, cnt = segment_crops(eng, crops, mean, std, 512, 1024, 32, paste=comp.paste_target(), origins=[(b[0], b[1]) for b in mine],
                                   want_masks=True)
            counts += torch.from_numpy(cnt.sum(0)).to(dev)
        if dist is not None:
            all_reduce(counts)
            all_reduce(comp.map, op=dist.ReduceOp.MAX)        # the one exchange: max-composite is associative
        sync()
        return comp, counts

    # first pass: allocates the workspaces and the pinned staging buffers (once per process, not once per slide)
    t0 = time.perf_counter()
    detect_leg()
    segment_leg()
    t_first = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    # timed passes: the median of three per leg (one pass is 10-30 ms: a single sample carried the box's jitter)
    t_det, t_sg = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        rows = detect_leg()
        sync()
        t_det.append(time.perf_counter() - t0)
        dets = [[float(v) for v in r.strip().split(",")[5:10]] for r in rows]
        merged_det = merge.merge_detections(dets, mpp, mpp, 0.35, 0.2) if dets else []
        t0 = time.perf_counter()
        comp, counts = segment_leg()
        t_sg.append(time.perf_counter() - t0)
    t_detect, t_seg = sorted(t_det)[1], sorted(t_sg)[1]
    t_total = time.perf_counter() - t_start
    # the same leg three times back to back (no detect / host merge in front of it: the GPU does not idle in between)
    t_b2b = []
    for _ in range(3):
        t0 = time.perf_counter()
        segment_leg()
        t_b2b.append(time.perf_counter() - t0)
    t_seg_b2b = sorted(t_b2b)[1]

    def mx(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    t_detect_m, t_seg_m, t_read_m, t_total_m, t_first_m = mx(t_detect), mx(t_seg), mx(t_read), mx(t_total), mx(t_first)
    t_seg_b2b_m = mx(t_seg_b2b)
    # every rank's ranges, for the record (and for the rehearsal tests: uneven and empty ranges must add up)
    ranges = torch.zeros((world, 4), dtype=torch.int64, device=dev)
    ranges[rank] = torch.tensor([lo, hi, blo, bhi], dtype=torch.int64)
    if dist is not None:
        all_reduce(ranges)
    if rank == 0:
        print(json.dumps({
            "config": "cfg 4: detect -> merge -> crop -> segment -> composite, one synthetic %d x %d slide, %d rank(s)" % (S, S, world),
            "data": "dry-run (CPU stand-ins, no device work)" if dry else "synthetic",
            "window_ranges": [[int(a), int(b)] for a, b, _, _ in ranges.tolist()], "crop_ranges": [[int(c), int(d)] for _, _, c, d in ranges.tolist()],
            "windows": len(wins), "window_px": [plan.window_y, plan.window_x], "crops": len(boxes_all),
            "detect_leg_s": round(t_detect_m, 4), "windows_per_s": round(len(wins) / t_detect_m, 1),
            "segment_composite_leg_s": round(t_seg_m, 4), "segment_composite_leg_back_to_back_s": round(t_seg_b2b_m, 4), "crops_per_s": round(len(boxes_all) / t_seg_m, 1),
            "gpu_legs_s": round(t_detect_m + t_seg_m, 4),
            "first_pass_gpu_legs_s": round(t_first_m, 3),
            "synthetic_region_generation_s": round(t_read_m, 3),
            "slide_total_s": round(t_total_m, 3),
            "detector_rows_rank0": len(rows), "detector_merged_rank0": len(merged_det),
            "pixel_totals": [int(v) for v in counts.tolist()], "map_nonzero": int((comp.map > 0).sum().item()),
            "note": "max over ranks per leg; legs = median of three timed passes over the slide after a first pass (first_pass_gpu_legs_s) that "
                    "allocates workspaces and pinned staging once per process; pageable numpy regions in, crop-size maps out; "
                    "the region generator stands in for OpenSlide and is CPU numpy",
        }))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
