"""Detect -> merge -> crop -> segment -> composite for one slide (BASELINE cfg 4), sharded by tile range.

The reference chains five CLI scripts through files on disk; this module runs the
same stages in one process per GPU:

  1. sliding-window detection over the slide (detect.scan_slide; the detector network is a plug-in),
  2. greedy merge of the overlapping windows' boxes (merge.merge_detections),
  3. level-0 crop of every merged box (make_seg_data.py:357-361) -> normalise + resize to the network size
     on the GPU (gs_crop_preprocess) -> ESPNet forward -> argmax -> nearest resize back to the crop size,
  4. max-composite of the crop masks on the 1/8-scale slide map + per-class pixel totals.

Ranks split stage 1 by window range and stage 3 by crop range (shard.rank_range); the only exchanges are
an all-gather of the few detection rows before the merge and the final map/count reduction.
"""
import numpy as np
import torch

from . import detect, merge
from .composite import SlideCompositor
from .shard import all_reduce_any, rank_range


def engine_classes(engine):
    """class count of an engine or of an ensemble (list of engines: the members agree, engine.segment_crops checks)"""
    e = engine[0] if isinstance(engine, (list, tuple)) else engine
    return int(e.classes)


def segment_crops(engine, crops, mean, std, net_h, net_w, batch=32, paste=None, origins=None, want_masks=True):
    """crops: list of uint8 BGR [h,w,3] arrays of any size -> (list of uint8 class maps [h,w], counts int64 [n,classes]) through
    the library's batched crop pipeline (gs_espnet_segment_crops_host); with `paste` (+ origins) the maps are also
    max-composited into the slide map on the GPU, in the same launches."""
    r = engine.segment_crops(crops, mean, std, net_h, net_w, batch, want_masks=want_masks, paste=paste, origins=origins)
    return r["masks"], r["counts"]


def run_slide(engine, read_region, slide_w, slide_h, mpp_x, mpp_y, detector, mean, std, window_um=2000, overlap=0.1,
              conf_threshold=0.2, overlap_threshold=0.35, objective_power=40, level_downsamples=(1.0, 2.0, 4.0, 8.0),
              net_h=512, net_w=1024, rank=0, world=1, dist=None, batch=32, detector_batch=1):
    """read_region(x, y, w, h, downsample) -> uint8 RGB [h,w,3] of the slide at that downsample (level-0 origin).
    Returns dict(boxes=merged boxes, masks=this rank's crop masks, map=1/8 class map (rank 0 / all ranks when
    dist is None), counts=per-class pixel totals over all crops: int64 [engine.classes])."""
    dev = engine.device
    level, ds = detect.pick_level(objective_power, level_downsamples)
    plan = detect.plan_windows(slide_w, slide_h, mpp_x, mpp_y, ds, window_um, overlap)
    rows = detect.scan_slide(lambda x, y, w, h: read_region(x, y, w, h, ds), detector, plan, conf_threshold, "site", "slide",
                             "slide.ndpi", rank=rank, world=world, batch=detector_batch)
    dets = [[float(v) for v in r.strip().split(',')[5:10]] for r in rows]
    if dist is not None and world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, dets)            # a few rows per rank, before the host merge
        dets = [d for part in gathered for d in part]
    boxes = merge.merge_detections(dets, mpp_x, mpp_y, overlap_threshold, conf_threshold)
    boxes = [[int(b[0]), int(b[1]), int(b[2]), int(b[3]), b[4]] for b in boxes]      # merged-CSV rounding (:121-124)
    boxes = [b for b in boxes if b[2] > b[0] and b[3] > b[1]]
    lo, hi = rank_range(len(boxes), rank, world)
    crops = []
    for b in boxes[lo:hi]:
        rgb = read_region(b[0], b[1], b[2] - b[0], b[3] - b[1], 1.0)               # make_seg_data.py:358
        crops.append(np.ascontiguousarray(rgb[:, :, ::-1]))                        # cv2.imread order: BGR
    comp = SlideCompositor(slide_w, slide_h, dev)
    masks, cnt = [], None
    if crops:     # resample, forward, resize back, count and paste: one pipeline call for the rank's crops
        masks, cnt = segment_crops(engine, crops, mean, std, net_h, net_w, batch, paste=comp.paste_target(),
                                   origins=[(b[0], b[1]) for b in boxes[lo:hi]])
    # one bin per class of the MODEL (VisualizeResults_iou.py:151-156 counts `args.classes` values, :315 passes them on)
    counts = torch.zeros(engine_classes(engine), dtype=torch.int64, device=dev)
    if cnt is not None:
        counts += torch.from_numpy(cnt.sum(0)).to(dev)
    if dist is not None and world > 1:
        all_reduce_any(counts, dist)                       # (backend-aware: device tensors as they are under RCCL, via the host under gloo)
        all_reduce_any(comp.map, dist, op=dist.ReduceOp.MAX)    # max-composite is associative: one collective
    return {"boxes": boxes, "masks": masks, "map": comp.map, "counts": counts, "plan": plan, "range": (lo, hi)}
