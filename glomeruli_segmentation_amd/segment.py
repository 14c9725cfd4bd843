#!/usr/bin/env python3
"""Per-patch glomerular segmentation driver: same flags and outputs as the reference's
module/espnet/test/VisualizeResults_iou.py (argparse :292-321, loop :84-241), with the batch-1
eager loop replaced by batched HIP passes.

    python -m glomeruli_segmentation_amd.segment --rgb_data_dir DIR --weights espnet_fold1.pth \
        --mean 204.60071 170.19359 199.57469 --std 20.61257 42.92207 28.401505 --gpu_id 0 ...

Additive flags: --batch (tiles per GPU pass), --imageData.  Started one process per GPU (torchrun / RANK, LOCAL_RANK,
WORLD_SIZE) every rank takes a contiguous range of the sorted crop list on its own GPU, and rank 0 writes the SAME summary
files a single process writes: CSV rows gathered in order, the confusion matrix summed over ranks.
"""
import base64
import glob
import io
import json
import os
import sys
import threading
import time
from argparse import ArgumentParser
from collections import defaultdict

import numpy as np

from . import imageops
from .shard import rank_range


def build_parser():
    p = ArgumentParser(description='Glomerular segmentation on the cropped images')
    p.add_argument('--rgb_data_dir', default="./data", required=True)
    p.add_argument('--label_data_dir', default=None)
    p.add_argument('--img_extn', default="PNG")
    p.add_argument('--inWidth', type=int, default=1024)
    p.add_argument('--inHeight', type=int, default=512)
    p.add_argument('--scaleIn', type=int, default=1)
    p.add_argument('--modelType', type=int, default=1, help='1=ESPNet, 2=ESPNet-C')
    p.add_argument('--savedir', default='./results')
    p.add_argument('--gpu_id', default=-1, type=int)
    p.add_argument('--decoder', action='store_true')
    p.add_argument('--weights', required=True)
    p.add_argument('--mean', required=True, nargs='*')
    p.add_argument('--std', required=True, nargs='*')
    p.add_argument('--p', default=2, type=int)
    p.add_argument('--q', default=8, type=int)
    p.add_argument('--cityFormat', action='store_true')
    p.add_argument('--colored', action='store_true')
    p.add_argument('--overlay', action='store_true')
    p.add_argument('--classes', default=5, type=int)
    p.add_argument('--batch', default=32, type=int, help='tiles per GPU pass (additive flag)')
    p.add_argument('--workers', default=None, type=int,
                   help='host threads that decode the next batch and encode / write the previous one while a batch is on the GPU '
                        '(additive flag; default: the CPUs of this process, at most 16; 0 = the serial loop)')
    p.add_argument('--imageData', default='orig', choices=['orig', 'classmap', 'none'],
                   help="what the JSON's imageData holds (additive flag): 'orig' = the original crop as the reference stores it "
                        "(VisualizeResults_iou.py:179), 'classmap' = the class map (the variant commented out at :178, which is "
                        "what eval_wsi_segmentation.py decodes), 'none' = null (the class map is always written beside the JSON)")
    return p


def load_state_dict_file(path):
    """torch .pth state_dict (the reference's format) or an .npz of the same keys."""
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    import torch
    return torch.load(path, map_location="cpu")


def confusion(pred, ref, classes):
    """iouEval.fast_hist (module/common/IOUEval.py:19-21): rows = ground truth, cols = prediction."""
    k = (ref >= 0) & (ref < classes)
    return np.bincount(classes * ref[k].astype(int) + pred[k].astype(int), minlength=classes ** 2).reshape(classes, classes)


def metric_right(hist):
    """iouEval.getMetricRight (IOUEval.py:63-69), epsilon included."""
    eps = 0.00000001
    overall = np.diag(hist).sum() / (hist.sum() + eps)
    per_acc = np.diag(hist) / (hist.sum(1) + eps)
    per_iu = np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist) + eps)
    return overall, per_acc, per_iu, np.nanmean(per_iu)


OVERLAY_WEIGHTS = (0.4, 0.6)      # cv2.addWeighted(img_orig, 0.4, classMap_numpy_color, 0.6, 0), VisualizeResults_iou.py:145


def segment_batch(engine, images, mean, std, width, height, batch, want_net_maps=False, want_overlay=False):
    """What the loop body needs from the GPU for one batch of crops (:107-129, :139-146, :151-155), as a dict:
    masks (crop-size class maps), net_maps (network-resolution maps | None), counts (int64 [n, classes]: pixels per class of the
    crop-size maps) and overlays (the palette-coloured map blended over the crop, BGR | None) -- counts and overlays come out of
    the batched crop pipeline with the maps (gs_espnet_segment_crops_host: crops_back_kernel counts, crops_overlay_kernel
    blends), the host only encodes.  ESPNet-C (modelType 2) has no such pipeline: its maps come from segment_images and the
    two by-products from _byproducts (on the GPU as well)."""
    if engine.encoder_only or not images:
        masks, net = segment_images(engine, images, mean, std, width, height, batch, want_net_maps=True)
        counts, overlays = _byproducts(engine, images, masks, want_overlay)
        return {"masks": masks, "net_maps": net if want_net_maps else None, "counts": counts, "overlays": overlays}
    r = engine.segment_crops(images, mean, std, height, width, batch, want_masks=True, want_net_maps=want_net_maps, want_hist=True,
                             overlay=(imageops.PALETTE, OVERLAY_WEIGHTS[0], OVERLAY_WEIGHTS[1]) if want_overlay else None)
    return {"masks": r["masks"], "net_maps": list(r["net_maps"]) if want_net_maps else None, "counts": r["counts"],
            "overlays": r["overlays"]}


def _byproducts(engine, images, masks, want_overlay):
    """counts (:151-155) and overlays (:139-146) of class maps that did not come out of the batched crop pipeline (ESPNet-C): on the
    engine's GPU when it has one -- torch.bincount and gs_overlay_classmap, the arithmetic of crops_overlay_kernel -- else (the CPU tests'
    stand-in engines) the same arithmetic in numpy"""
    classes = engine.classes
    dev = getattr(engine, "device", None)
    if dev is None or not hasattr(engine, "lib"):
        counts = np.array([np.bincount(np.asarray(m).ravel(), minlength=classes)[:classes] for m in masks], dtype=np.int64).reshape(len(masks), classes)
        overlays = [imageops.add_weighted(im, OVERLAY_WEIGHTS[0], imageops.colourise(m), OVERLAY_WEIGHTS[1])
                    for im, m in zip(images, masks)] if want_overlay else None
        return counts, overlays
    import ctypes
    import torch
    from . import _lib
    pal = torch.from_numpy(np.ascontiguousarray(imageops.PALETTE)).to(dev)
    counts = np.zeros((len(masks), classes), dtype=np.int64)
    overlays = [] if want_overlay else None
    with torch.cuda.device(dev):
        for i, (im, m) in enumerate(zip(images, masks)):
            mg = torch.from_numpy(np.ascontiguousarray(m)).to(dev)
            counts[i] = torch.bincount(mg.flatten().long(), minlength=classes)[:classes].cpu().numpy()
            if want_overlay:
                ig = torch.from_numpy(np.ascontiguousarray(im)).to(dev)
                out = torch.empty_like(ig)
                _lib.check(engine.lib.gs_overlay_classmap(ig.data_ptr(), mg.data_ptr(), int(m.shape[0]), int(m.shape[1]), pal.data_ptr(),
                                                          int(pal.shape[0]), ctypes.c_float(OVERLAY_WEIGHTS[0]), ctypes.c_float(OVERLAY_WEIGHTS[1]),
                                                          out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
                overlays.append(out.cpu().numpy())
    return counts, overlays


def segment_images(engine, images, mean, std, width, height, batch, want_net_maps=False):
    """images: list of HxWx3 uint8 BGR crops of any size -> list of class maps at crop size
    (with want_net_maps: (crop-size maps, network-resolution maps) -- the reference scores the latter, :202).
    A list of network-sized tiles goes down the fused uint8 pipeline (gs_espnet_segment_host); any other list goes through
    the batched crop pipeline (gs_espnet_segment_crops_host), which normalises and resizes on the GPU exactly in the
    reference's order (:107-116) for a whole batch per launch.  With an encoder-only engine (modelType 2) every crop is
    resampled on its own and the 1/8-scale logits are upsampled x8 bilinearly as the reference's `up` module does
    (:259-261,125-126)."""
    import torch
    out = [None] * len(images)
    net = [None] * len(images)
    enc = engine.encoder_only
    if not enc and images:
        if all(im.shape[:2] == (height, width) for im in images):
            # network-sized tiles: the fused uint8 path (normalisation through the stem's table), pinned pipeline
            masks, _ = engine.segment_host(np.stack(images), mean, std, batch=batch, want_hist=False)
            out = list(masks)
            net = list(masks)
        else:
            # any sizes: ONE call of the batched crop pipeline (resample -> forward -> argmax -> resize back per batch)
            r = engine.segment_crops(images, mean, std, height, width, batch, want_masks=True, want_net_maps=want_net_maps,
                                     want_hist=False)
            out = r["masks"]
            if want_net_maps:
                net = list(r["net_maps"])
        return (out, net) if want_net_maps else out
    from .engine import crop_preprocess, mask_resize_nearest
    for s in range(0, len(images), batch):
        idx = list(range(s, min(s + batch, len(images))))
        x = torch.empty((len(idx), 3, height, width), dtype=torch.float32, device=engine.device)
        for j, i in enumerate(idx):      # crop stage on the GPU: normalise + bilinear resize fused (:107-116)
            crop_preprocess(torch.from_numpy(images[i]).to(engine.device), mean, std, height, width, out=x[j])
        logits = engine.forward_logits(x)
        logits = torch.nn.functional.interpolate(logits, scale_factor=8, mode="bilinear", align_corners=False)
        cls = logits.max(1)[1].byte()       # :128
        cls_host = cls.cpu().numpy() if want_net_maps else None
        for j, i in enumerate(idx):
            h, w = images[i].shape[:2]
            out[i] = mask_resize_nearest(cls[j], h, w).cpu().numpy()    # :129
            if want_net_maps:
                net[i] = cls_host[j]
    return (out, net) if want_net_maps else out


def img_arr_to_b64(arr):
    """labelme.utils.img_arr_to_b64 (labelme 3.16): the array as a PNG, base64.  The reference hands it the cv2 (BGR) crop,
    which PIL takes for RGB: the stored picture has its channels swapped, and so has this one."""
    from PIL import Image
    f = io.BytesIO()
    Image.fromarray(arr).save(f, format='PNG')
    return base64.b64encode(f.getvalue()).decode('utf-8')


def _load_crop(name, label_name):
    """decode stage (worker thread), one crop: the image as cv2.imread gives it (:103) and its label image (:191-192)"""
    from PIL import Image
    with _timed("decode_png"):
        return imageops.imread_bgr(name), (None if label_name is None else np.asarray(Image.open(label_name)))


class _Done:
    """a finished "future": the serial path (--workers 0) makes its calls on the spot"""

    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


def _timed(stage):
    """per-stage host timing of _emit_crop / _load_crop for tools/bench_cli.py: STAGE_SECONDS[stage] accumulates thread seconds
    when it is a dict, and costs one `is None` test otherwise"""
    class _T:
        def __enter__(self):
            self.t0 = time.perf_counter() if STAGE_SECONDS is not None else 0.0

        def __exit__(self, *exc):
            if STAGE_SECONDS is not None:
                dt = time.perf_counter() - self.t0
                with _STAGE_LOCK:
                    STAGE_SECONDS[stage] = STAGE_SECONDS.get(stage, 0.0) + dt
            return False
    return _T()


STAGE_SECONDS = None          # tools/bench_cli.py sets a dict here to collect the breakdown
_STAGE_LOCK = threading.Lock()


def _emit_crop(args, img_name, label_name, img, cmap, net_map, lab, lab_r, counts, overlayed):
    """Everything the loop body writes for ONE crop after the forward (:131-231): overlay / original images, the counts row,
    the class map, the labelme JSON, with a label the per-image accuracy row and the combined image.  Host work only -- PNG /
    JPEG encoding, contour tracing, base64 -- and no state shared with other crops, so it runs on a worker thread while the
    GPU is busy with the next batch.  `counts` (pixels per class of the crop-size map, :151-155) and `overlayed` (:139-146; None
    when neither --colored nor a label asks for it) arrive from the GPU pass.  Returns what the summary files need: (pixel row,
    accuracy row | None, (patient, label values) | None, confusion matrix | None)."""
    from PIL import Image
    from .contours import labelme_dict
    patient = os.path.basename(os.path.dirname(img_name))
    name = os.path.basename(img_name)
    stem = name.rsplit(".", 1)[0]
    odir = os.path.join(args.savedir, patient)
    os.makedirs(odir, exist_ok=True)
    if args.colored and args.overlay:                                                      # :139-148 (the blend itself: the GPU pass)
        with _timed("overlay_jpeg"):
            imageops.imwrite_bgr(os.path.join(odir, stem + "_overlay.jpg"), overlayed)
        with _timed("org_png"):
            imageops.imwrite_bgr(os.path.join(odir, stem + "_org.png"), img)
    # :151-155: the reference's row names the five classes of its networks; a model with fewer has zeros there
    c5 = [int(counts[c]) if c < len(counts) else 0 for c in range(5)]
    row_pixel = "{},{},{},{},{},{},{}\n".format(patient, name.replace(args.img_extn, 'png'), *c5)
    out_map = imageops.relabel_city(cmap) if args.cityFormat else cmap                     # :158-159
    # The class map is always written beside the JSON (additive: the WSI compositor takes it from there); what the
    # JSON's imageData holds follows --imageData (the reference: the ORIGINAL crop, :179, although
    # eval_wsi_segmentation.py decodes it as a class map -- SURVEY quirks).
    with _timed("classmap_png"):
        Image.fromarray(out_map).save(os.path.join(odir, stem + "_classmap.png"))
    with _timed("contours"):
        body = labelme_dict(out_map, name, stem + "_classmap.png")                         # :161-177
    with _timed("base64_png"):
        body["imageData"] = (img_arr_to_b64(img) if args.imageData == 'orig' else
                             img_arr_to_b64(np.ascontiguousarray(out_map, dtype=np.uint8)) if args.imageData == 'classmap' else None)
    with _timed("json"):
        with open(os.path.join(odir, name.replace(args.img_extn, 'json')), 'w') as f:
            json.dump(body, f, indent=4)
    if label_name is None:
        return row_pixel, None, None, None
    # the reference scores at network resolution (:195-203): the label is nearest-resized to the network size (lab_r, made on
    # the GPU by the caller) and compared with img_out.max(1)[1] itself, NOT with the map that went to crop size and back
    hist = confusion(net_map.ravel(), lab_r.ravel(), args.classes)
    uniq = np.unique(lab_r)
    _, _, per_iu, _ = metric_right(hist)
    union = hist.sum(1) + hist.sum(0) - np.diag(hist)
    miou_each = np.nanmean(np.diag(hist)[uniq] / union[uniq])                               # :208-209
    flags = [1 if (uniq == c).any() else 0 for c in range(1, 5)]
    row_acc = "{}/{},{},{},{},{},{},{},{},{},{},{}\n".format(
        patient, name.replace(args.img_extn, 'png'), *flags, *per_iu[:5], miou_each)
    # original | ground truth overlay | prediction overlay (:215-231)
    gt_colour = imageops.colourise(np.minimum(lab, len(imageops.PALETTE) - 1).astype(np.uint8))
    combined = np.concatenate([img, imageops.add_weighted(img, 0.4, gt_colour, 0.6), overlayed], axis=1)
    cdir = os.path.join(args.savedir, "combined_images", patient)
    os.makedirs(cdir, exist_ok=True)
    imageops.imwrite_bgr(os.path.join(cdir, name.replace(args.img_extn, 'png')), combined)
    return row_pixel, row_acc, (patient, uniq.tolist()), hist


def default_workers():
    """host threads of the decode-ahead / write-behind pool: the CPUs this process may run on (its rank's share of the node,
    launch.place_rank), at most 16"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def evaluate(args, engine, rgb_list, label_list, rank=0, world=1, dist=None):
    """evaluateModel (:84-241) over this rank's range; the summary files are written by rank 0 for all ranks.

    The reference decodes, infers and writes one crop at a time on one thread (:100-231).  Here the GPU takes a batch per
    pass, and with args.workers > 0 the host work either side of it overlaps with it: a thread pool decodes the PNGs of batch
    k+1 and encodes / traces / writes the outputs of batch k-1 while batch k is on the GPU (PIL's codecs and the contour
    tracer release the interpreter lock).  Rows and matrices are collected in list order, so every output file is byte for
    byte what the serial path (workers == 0: the same two functions, called inline) writes."""
    from concurrent.futures import ThreadPoolExecutor
    from .shard import gather_rows, reduce_sum_to_all
    mean = [float(v) for v in args.mean]
    std = [float(v) for v in args.std]
    os.makedirs(args.savedir, exist_ok=True)
    total_hist = np.zeros((args.classes, args.classes), dtype=np.int64)
    rows_pixel, rows_acc, seen = [], [], []
    workers = max(0, int(getattr(args, "workers", 0) or 0))
    pool = ThreadPoolExecutor(max_workers=workers) if workers > 0 else None

    def run(fn, *a):
        return _Done(fn(*a)) if pool is None else pool.submit(fn, *a)

    def load(s):          # one future per crop: the decode of a batch spreads over the pool like its outputs do
        return [run(_load_crop, n_, l_) for n_, l_ in zip(rgb_list[s:s + args.batch], label_list[s:s + args.batch])]

    def collect(futs):
        nonlocal total_hist
        for f in futs:
            rp, ra, sn, hist = f.result()
            rows_pixel.append(rp)
            if ra is not None:
                rows_acc.append(ra)
                seen.append(sn)
                total_hist += hist

    starts = list(range(0, len(rgb_list), args.batch))
    try:
        nxt = load(starts[0]) if starts else None
        pending = []          # emit futures of the batches behind the one on the GPU, oldest first
        for bi, s in enumerate(starts):
            names, label_names = rgb_list[s:s + args.batch], label_list[s:s + args.batch]
            loaded = [f.result() for f in nxt]
            images, labels = [im for im, _ in loaded], [lb for _, lb in loaded]
            if bi + 1 < len(starts):      # decode ahead
                nxt = load(starts[bi + 1])
            want_overlay = bool(args.colored or any(l_ is not None for l_ in label_names))      # (:139, :215-231)
            t_gpu = time.perf_counter()
            r = segment_batch(engine, images, mean, std, args.inWidth, args.inHeight, args.batch, want_net_maps=True,
                              want_overlay=want_overlay)
            if STAGE_SECONDS is not None:
                with _STAGE_LOCK:
                    STAGE_SECONDS["gpu_pass"] = STAGE_SECONDS.get("gpu_pass", 0.0) + time.perf_counter() - t_gpu
            masks, net_maps, counts = r["masks"], r["net_maps"], r["counts"]
            overlays = r["overlays"] if want_overlay else [None] * len(images)
            labs_r = []
            for img_name, label_name, img, lab in zip(names, label_names, images, labels):
                if label_name is None:
                    labs_r.append(None)
                    continue
                assert os.path.basename(img_name) == os.path.basename(label_name)
                assert lab.shape[:2] == img.shape[:2]
                if lab.shape[:2] == (args.inHeight, args.inWidth):
                    labs_r.append(lab)
                else:
                    import torch
                    from .engine import mask_resize_nearest
                    labs_r.append(mask_resize_nearest(torch.from_numpy(np.array(lab, dtype=np.uint8)).to(engine.device),
                                                      args.inHeight, args.inWidth).cpu().numpy())      # :195 cv2.resize INTER_NEAREST
            futs = [run(_emit_crop, args, n_, l_, im, np.array(cm), np.array(nm) if nm is not None else None, lb, lr,
                        [int(v) for v in cn], None if ov is None else np.array(ov))
                    for n_, l_, im, cm, nm, lb, lr, cn, ov in zip(names, label_names, images, masks, net_maps, labels, labs_r, counts, overlays)]
            # (np.array: the maps and overlays are views of the pipeline's pinned output buffers, which the next call may reuse)
            pending.append(futs)
            if len(pending) > 2:          # write behind: at most two batches of outputs in flight
                collect(pending.pop(0))
        for futs in pending:
            collect(futs)
    finally:
        if pool is not None:
            pool.shutdown(wait=True)
    # ---- one set of summary files (:91-98,232-241): rows in the order of the sorted list, one confusion matrix
    rows_pixel = gather_rows(rows_pixel, rank, world, dist)
    rows_acc = gather_rows(rows_acc, rank, world, dist)
    seen = gather_rows(seen, rank, world, dist)
    total_hist = reduce_sum_to_all(total_hist, dist, getattr(engine, "device", None))
    have_labels = label_list and label_list[0] is not None
    if world > 1:
        have_labels = bool(reduce_sum_to_all(np.array([1 if have_labels else 0]), dist, getattr(engine, "device", None))[0])
    if rank != 0:
        return total_hist
    with open(os.path.join(args.savedir, "summary_pixel.csv"), "w") as f:
        f.write("patient_id, filename, background, glomerulus, crescent, sclerosis, mesangium\n")
        f.writelines(rows_pixel)
    with open(os.path.join(args.savedir, "summary_accuracy.csv"), "w") as f:
        f.write("filename,glomerulus, crescent, sclerosis, mesangium, background iou,glomerulus iou,crescent iou,"
                "sclerosis iou, mesangium iou,mIoU\n")
        f.writelines(rows_acc)
    with open(os.path.join(args.savedir, "summary_dataset.csv"), "w") as f:
        f.write("patient_id, glomerulus, crescent, sclerosis, mesangium\n")
        if have_labels:
            dataset_d = defaultdict(lambda: defaultdict(int))
            for patient, values in seen:
                for v in values:
                    dataset_d[patient][v] += 1
            for patient, vals in dataset_d.items():
                f.write(patient + "".join(",{}".format(vals[i]) for i in range(1, args.classes)) + "\n")
    if have_labels:
        o, pa, pi, m = metric_right(total_hist)
        with open(os.path.join(args.savedir, "overall_accuracy.txt"), "w") as f:
            f.write("overall_acc:{}, per_class_acc:{}, per_class_iou:{}, mIOU:{}".format(o, pa, pi, m))
    return total_hist


def main(argv=None):
    args = build_parser().parse_args(argv)
    print(args.decoder)
    if args.overlay:
        args.colored = True
    rgb_list = sorted(glob.glob(args.rgb_data_dir + "/*/*.PNG"))
    if args.label_data_dir is not None:
        label_list = sorted(glob.glob(args.label_data_dir + "/*/*.PNG"))
        assert len(rgb_list) == len(label_list)
    else:
        label_list = [None] * len(rgb_list)
    if args.modelType not in (1, 2):
        print('Model not supported')
        return 1
    if not os.path.isfile(args.weights):
        print('Pre-trained model file does not exist. Please check the --weights path')
        return -1
    if args.gpu_id < 0:
        # the reference's default: --gpu_id -1 = CPU (VisualizeResults_iou.py:252-255,302).  This build has no CPU product path
        # -- a silent fallback would be a different program answering under the same name -- so the default is refused, loudly
        print("--gpu_id %d selects the CPU in the reference (its default); this build runs the ESPNet forward on a HIP device "
              "only and has no CPU path: pass --gpu_id >= 0" % args.gpu_id, file=sys.stderr)
        return 2
    import torch
    from .engine import EspnetEngine
    from .shard import init_from_env
    rank, world, local, dist = init_from_env()
    if args.workers is None:          # (after init_from_env: a rank's CPU share is set there)
        args.workers = default_workers()
    # one process per GPU: every rank on its own device (LOCAL_RANK); a single process keeps the reference's --gpu_id
    dev_id = args.gpu_id if world == 1 or os.environ.get("GLOMSEG_ONE_GPU") == "1" else local
    torch.cuda.set_device(dev_id)
    print('cuda:{}'.format(dev_id))
    lo, hi = rank_range(len(rgb_list), rank, world)
    rgb_list, label_list = rgb_list[lo:hi], label_list[lo:hi]
    print("num of image:{}".format(len(rgb_list)))
    sd = load_state_dict_file(args.weights)
    if args.modelType == 2:
        # ESPNet-C (:267-272): the checkpoint holds the encoder's own keys; a full-network checkpoint is accepted too
        if any(k.startswith("encoder.") for k in sd):
            sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        engine = EspnetEngine(sd, classes=args.classes, p=args.p, q=args.q, encoder_only=True)
    else:
        engine = EspnetEngine(sd, classes=args.classes, p=args.p, q=args.q, lanes=2)
    from .shard import abort_rank, finish_ranks
    try:
        evaluate(args, engine, rgb_list, label_list, rank, world, dist)
    except BaseException:
        # a rank that fails alone must not wait for its peers in a barrier (shard.abort_rank) -- and it leaves FIRST: if the
        # failure was a GPU fault or a hang, closing the engine (a device synchronise) could block for ever with the peers
        # sitting in their gather; the process is exiting and the driver reclaims its device memory
        abort_rank(dist)
        engine.close()        # single process: tidy up, then let the exception (or a SystemExit's own code) through
        raise
    engine.close()
    finish_ranks(dist)
    return 0


if __name__ == '__main__':
    sys.exit(main())
