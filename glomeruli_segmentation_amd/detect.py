"""Host side of the sliding-window glomerulus detector: window geometry, level choice, box
post-processing and the detections CSV -- the call surface of the reference's
module/faster-rcnn/detect_glomus_test.py around its ``sess.run`` (:350-352).

The detector NETWORK is an external TensorFlow-1.12 frozen graph that is not part of the reference
(:419-427; download in example/README.md:22), so here it is a plug-in: any callable with the
``detect_box`` tensor contract

    detector(uint8 RGB [1,H,W,3]) -> (boxes [1,K,4] normalised [ymin,xmin,ymax,xmax],
                                      scores [1,K] descending, classes [1,K], num [1])

(:443-450).  Everything in this file is pinned by known-answer tests written from the reference's
formulas; parity for a network behind the plug-in is unpinned.
"""
import datetime
import math
import os
import sys
import time
from argparse import ArgumentParser
from dataclasses import dataclass

import numpy as np

from .shard import rank_range

STAINING_DIRS = {'OPT_PAS': '02_PAS', 'OPT_PAM': '03_PAM', 'OPT_MT': '05_MT', 'OPT_Azan': '06_Azan'}   # glomus_handler.py:46-60
STAINING_TYPES = ('OPT_PAM', 'OPT_MT', 'OPT_PAS', 'OPT_HE', 'OPT_Azan')                                 # glomus_handler.py:22-40
NDPI_EXT = ('.ndpi',)
PNG_EXT = ('.PNG', '.png')


def staining_dir(data_category):
    """GlomusHandler.get_staining_type (glomus_handler.py:46-60): '' for anything it does not list (OPT_HE included)."""
    return STAINING_DIRS.get(data_category, '')


def staining_type(data_category):
    """GlomusHandler.set_type (glomus_handler.py:22-40): the file-name prefix; unknown categories raise."""
    if data_category not in STAINING_TYPES:
        raise ValueError('Unknown Argument is given.:' + str(data_category))
    return data_category

DEFAULT_WINDOW_UM = 500      # detect_glomus_test.py:52-54
DEFAULT_OVERLAP = 0.5


def pick_level(objective_power, level_downsamples):
    """First pyramid level whose magnification is <= 5x; level 3 / downsample 8 when none is
    (detect_glomus_test.py:255-261)."""
    for level, ds in enumerate(level_downsamples):
        if objective_power / ds <= 5.0:
            return level, float(ds)
    return 3, 8.0


@dataclass
class WindowPlan:
    window_x_org: float      # window edge in level-0 pixels
    window_y_org: float
    x_split_times: int
    y_split_times: int
    window_x: int            # window edge in pixels of the level that is read
    window_y: int
    step_x: int              # window stride
    step_y: int
    downsample: float

    def origins(self):
        """Row-major (i, j, x_start, y_start) of every window (:270-273)."""
        return [(i, j, self.step_x * i, self.step_y * j)
                for j in range(self.y_split_times) for i in range(self.x_split_times)]


def plan_windows(width, height, mpp_x, mpp_y, downsample, window_um=None, overlap=None, from_image=False):
    """calc_window_size (:286-304) + the stride of scan_region (:266-268; level-0 pixels) or of
    scan_region_from_image (:218-219; pixels of the down-sampled PNG) when ``from_image``."""
    if window_um is None or window_um == '':
        window_um, overlap = DEFAULT_WINDOW_UM, DEFAULT_OVERLAP
    wx_org = float(window_um) / mpp_x
    wy_org = float(window_um) / mpp_y
    xs = int(math.ceil(width / wx_org / (1.0 - overlap)))
    ys = int(math.ceil(height / wy_org / (1.0 - overlap)))
    wx = int(math.ceil(wx_org / downsample))
    wy = int(math.ceil(wy_org / downsample))
    if from_image:
        sx, sy = int(wx * (1.0 - overlap)), int(wy * (1.0 - overlap))
    else:
        sx, sy = int(wx_org * (1.0 - overlap)), int(wy_org * (1.0 - overlap))
    return WindowPlan(wx_org, wy_org, xs, ys, wx, wy, sx, sy, float(downsample))


def boxes_from_detector(boxes, scores, window_x, window_y, thresh=0.5):
    """Threshold + denormalise (:355-368).  Quirk kept: the first len(score >= thresh) boxes are
    taken by position, which equals selection by index only because detector outputs are sorted by
    descending score.
    The products are formed in float64: the reference runs on the NumPy 1.x of its TensorFlow-1.12 stack, where
    `WINDOW_X * xmin` (Python int x float32 scalar) promotes to float64; under NumPy 2 the same expression stays in
    float32 and int() of it can land one pixel away."""
    boxes = np.squeeze(np.asarray(boxes))
    score = np.squeeze(np.asarray(scores))
    boxes = boxes.reshape(-1, 4)
    score = score.reshape(-1)
    n = int(np.count_nonzero(score >= thresh))
    if n == 0:
        return []
    b = boxes[:n].astype(np.float64)
    px = np.stack([window_x * b[:, 1], window_y * b[:, 0], window_x * b[:, 3], window_y * b[:, 2]], 1).astype(np.int64)   # int(): towards zero
    return [[x1, y1, x2, y2, sc] for (x1, y1, x2, y2), sc in zip(px.tolist(), score[:n])]


def csv_rows(bs, x_start, y_start, downsample, site_name, specimen_id, file_name, now=None):
    """Window -> level-0 coordinates and the detections CSV row format (:306-326)."""
    rows = []
    stamp = (now or datetime.datetime.today()).strftime('%Y-%m-%dT%H:%M:%S')
    for b in bs:
        if b[4] > 0:
            rows.append('"' + site_name + '","' + specimen_id + '","' + file_name + '",new,' + stamp + ','
                        + str(x_start + (b[0] * downsample)) + ',' + str(y_start + (b[1] * downsample)) + ','
                        + str(x_start + (b[2] * downsample)) + ',' + str(y_start + (b[3] * downsample)) + ','
                        + str(b[4]) + '\n')
    return rows


def scan_slide(read_region, detector, plan, conf_threshold, site_name, specimen_id, file_name, rank=0, world=1,
               from_image=False, now=None, batch=1):
    """The HOT LOOP of scan_region (:270-284) for this rank's contiguous window range.
    read_region(x_start, y_start, window_x, window_y) -> uint8 RGB [window_y, window_x, 3].
    batch > 1 hands the detector that many windows per call ([B,H,W,3] in, [B,K,4] / [B,K] / [B,K] / [B] out: what
    FrcnnDetector takes); the reference's own loop is batch 1."""
    wins = plan.origins()
    lo, hi = rank_range(len(wins), rank, world)
    rows = []
    mine = wins[lo:hi]
    # a detector with a host pipeline (FrcnnDetector.detect_host: pinned uploads one batch ahead of the forward) is handed
    # several batches of windows per call; any other callable gets `batch` windows per call
    host = getattr(detector, "detect_host", None) if batch > 1 else None
    per_call = max(batch, 1) * (4 if host is not None else 1)
    for s in range(0, len(mine), per_call):
        chunk = mine[s:s + per_call]
        ims = []
        for i, j, xs, ys in chunk:
            im = np.asarray(read_region(xs, ys, plan.window_x, plan.window_y))
            if im.shape[-1] == 4:
                im = im[:, :, :3]                        # drop alpha (:277-278)
            ims.append(im)
        if host is not None:
            boxes, scores, classes, num = host(ims, batch=batch)
        else:
            boxes, scores, classes, num = detector(np.stack(ims))
        boxes, scores = np.asarray(boxes), np.asarray(scores)
        for k, (i, j, xs, ys) in enumerate(chunk):
            bs = boxes_from_detector(boxes[k], scores[k], plan.window_x, plan.window_y, conf_threshold)
            x0, y0 = (xs * plan.downsample, ys * plan.downsample) if from_image else (xs, ys)     # :234 vs :283
            rows.extend(csv_rows(bs, x0, y0, plan.downsample, site_name, specimen_id, file_name, now))
    return rows


def parse_target_line(line):
    """One line of the target list: "id/file[,w,h,power,ds,mppx,mppy]"; short lines zero the slide
    metadata silently, as the reference does (:112-129).  Returns None for '#' comment lines."""
    parts = line.strip().split(',')
    meta = dict(width=0, height=0, objective_power=0.0, downsample=0.0, mpp_x=0.0, mpp_y=0.0)
    if len(parts) >= 7:
        meta = dict(width=int(parts[1]), height=int(parts[2]), objective_power=float(parts[3]),
                    downsample=float(parts[4]), mpp_x=float(parts[5]), mpp_y=float(parts[6]))
    ids = parts[0].split('/')
    if ids[0].startswith('#'):
        return None
    meta["specimen_id"] = ids[0]
    meta["file_name"] = ids[1] if len(ids) > 1 else ""
    return meta


def find_slide_file(data_dir, staining, specimen_id, file_name):
    """The candidate search of split_all (:131-148): the first directory entry (in os.listdir order) whose stem occurs in
    `file_name` and whose extension is a slide extension is processed, then the loop breaks.
    Returns (entry, 'ndpi' | 'png') or (None, None)."""
    target = os.path.join(data_dir, staining, specimen_id)
    if not os.path.isdir(target):
        return None, None
    for candidate in os.listdir(target):
        body, ext = os.path.splitext(candidate)
        if file_name.find(body) >= 0 and ext in NDPI_EXT:
            return candidate, 'ndpi'
        if file_name.find(body) >= 0 and ext in PNG_EXT:
            return candidate, 'png'
    return None, None


def scan_png(path, detector, meta, window_um, overlap, conf_threshold, site_name, rank=0, world=1, batch=16, now=None, log=None):
    """split() for a PNG slide (:170-173) + scan_region_from_image (:196-234): the PNG is the slide at
    meta['downsample']; windows are PIL crops (black beyond the image, as Image.crop pads), rows are lifted to level 0."""
    from PIL import Image
    with Image.open(path) as img:
        plan = plan_windows(meta["width"], meta["height"], meta["mpp_x"], meta["mpp_y"], meta["downsample"], window_um, overlap,
                            from_image=True)

        def read(xs, ys, w, h):
            return np.asarray(img.crop((xs, ys, xs + w, ys + h)))
        rows = scan_slide(read, detector, plan, conf_threshold, site_name, meta["specimen_id"], os.path.basename(path), rank=rank,
                          world=world, from_image=True, now=now, batch=batch)
    return rows


def load_detector(model, model_name, synthetic_seed=None):
    """--model / --model_name -> a detector callable.  The reference joins model/model/model_name and parses a TensorFlow
    frozen graph (:413-427); that graph is an external download and TensorFlow is absent here, so this build takes the
    weights of its own assembled detector (detector.FrcnnDetector) as an .npz of the tensors detector.LAYERS names."""
    from .detector import FrcnnDetector, synthetic_weights
    if synthetic_seed is not None:
        return FrcnnDetector(synthetic_weights(synthetic_seed))
    cands = []
    if model:
        cands = [os.path.join(model, model, model_name), os.path.join(model, model_name), model]
    path = next((c for c in cands if os.path.isfile(c)), None)
    if path is None:
        raise FileNotFoundError("no detector weight file under --model %r (--model_name %r)" % (model, model_name))
    if not path.endswith(".npz"):
        raise ValueError("%s: this build cannot load a TensorFlow frozen graph (the reference's detector network is an external "
                         "download, and TensorFlow is not part of this stack); pass an .npz with the tensors of "
                         "glomeruli_segmentation_amd.detector.LAYERS" % path)
    z = np.load(path)
    return FrcnnDetector({k: z[k] for k in z.files})


def split_all(args, detector, rank=0, world=1, dist=None, out=sys.stdout):
    """GlomusDetector.__init__ output paths (:70-84) + split_all (:93-159): every slide of the target list, rows to
    <TYPE><output_file_ext>.csv, wall time per slide to ..._log.csv.  Ranks shard the windows of every slide; rank 0 writes."""
    from .shard import gather_rows
    type_name = staining_type(args.data_category)
    sdir = staining_dir(args.data_category)
    window_um, overlap = args.window_size, args.overlap_ratio
    if window_um is None or window_um == '':
        window_um, overlap = DEFAULT_WINDOW_UM, DEFAULT_OVERLAP
    site_name = args.data_dir.split('/')[-2]                       # :103-104
    out_path = os.path.join(args.output_dir, type_name + args.output_file_ext + '.csv')
    log_path = os.path.join(args.output_dir, type_name + args.output_file_ext + '_log.csv')
    if rank == 0:
        os.makedirs(args.output_dir, exist_ok=True)
    if not os.path.isfile(args.target_list):                        # :107: nothing is read; the output file exists, empty
        if rank == 0:
            open(out_path, "w").close()
        return out_path
    with open(args.target_list, 'r') as f:
        lines = f.readlines()
    out_file = open(out_path, "w") if rank == 0 else None
    log_file = open(log_path, "w") if rank == 0 else None
    if log_file:
        log_file.write('file,time\n')
    try:
        for line in lines:
            meta = parse_target_line(line)
            if meta is None:
                continue
            entry, kind = find_slide_file(args.data_dir, sdir, meta["specimen_id"], meta["file_name"])
            if kind is None:
                continue
            t0 = time.time()
            path = os.path.join(args.data_dir, sdir, meta["specimen_id"], entry)
            if kind == 'png':
                rows = scan_png(path, detector, meta, window_um, overlap, args.conf_threshold, site_name, rank, world, args.batch)
            else:
                rows = scan_ndpi(path, detector, meta, window_um, overlap, args.conf_threshold, site_name, rank, world, args.batch)
            rows = gather_rows(rows, rank, world, dist)
            if rank == 0:
                out_file.writelines(rows)
                out_file.flush()
                log_file.write('"{}",{}\n'.format(meta["file_name"], time.time() - t0))
                log_file.flush()
                print('{}: {} boxes at or above the threshold'.format(entry, len(rows)), file=out)
    finally:
        if out_file:
            out_file.close()
        if log_file:
            log_file.close()
    return out_path


def scan_ndpi(path, detector, meta, window_um, overlap, conf_threshold, site_name, rank=0, world=1, batch=16):
    """split() for an .ndpi slide (:174-185) + scan_region (:236-284).  Needs OpenSlide, which this image does not have:
    slide decoding is outside the rebuilt path (out of scope) -- the call fails loudly without it."""
    try:
        import openslide
    except ImportError as e:
        raise RuntimeError("%s: reading .ndpi slides needs the openslide package, which is not installed; convert the slide to a "
                           "PNG at the scan level and list its metadata in the target list (the reference's PNG branch)" % path) from e
    with openslide.open_slide(path) as slide:
        width, height = slide.dimensions
        mpp_x = float(slide.properties[openslide.PROPERTY_NAME_MPP_X])
        mpp_y = float(slide.properties[openslide.PROPERTY_NAME_MPP_Y])
        power = int(slide.properties[openslide.PROPERTY_NAME_OBJECTIVE_POWER])
        level, ds = pick_level(power, slide.level_downsamples)
        plan = plan_windows(width, height, mpp_x, mpp_y, ds, window_um, overlap)

        def read(xs, ys, w, h):
            return np.asarray(slide.read_region((xs, ys), level, (w, h)))
        return scan_slide(read, detector, plan, conf_threshold, site_name, meta["specimen_id"], os.path.basename(path), rank=rank,
                          world=world, batch=batch)


def build_parser():
    """argparse surface of detect_glomus_test.py:385-405 (+ additive flags, listed last)."""
    p = ArgumentParser(description='Load RoI')
    p.add_argument('--model', dest='model', type=str)
    p.add_argument('--target_list', dest='target_list', type=str)
    p.add_argument('--data_dir', dest='data_dir', type=str)
    p.add_argument('--staining', dest='data_category', type=str, default='OPT_PAM')
    p.add_argument('--output_dir', dest='output_dir', type=str, default='./output')
    p.add_argument('--output_file_ext', dest='output_file_ext', type=str, default='_GlomusList')
    p.add_argument('--window_size', dest='window_size', type=int)
    p.add_argument('--overlap_ratio', dest='overlap_ratio', type=float)
    p.add_argument('--conf_threshold', dest='conf_threshold', type=float, default=0.6)
    p.add_argument('--model_name', dest='model_name', default="frozen_inference_graph.pb", type=str)
    # additive
    p.add_argument('--batch', dest='batch', type=int, default=16, help='windows per detector forward (the reference: 1)')
    p.add_argument('--gpu_id', dest='gpu_id', type=int, default=0, help='HIP device of a single-process run (ranks use LOCAL_RANK)')
    p.add_argument('--synthetic_weights', dest='synthetic_weights', type=int, default=None,
                   help='seeded synthetic detector weights instead of --model (tests, benchmarks: no trained weights exist offline)')
    return p


def main(argv=None, detector=None):
    """`python -m glomeruli_segmentation_amd.detect ...`: the __main__ of detect_glomus_test.py:408-456 with the TensorFlow
    session replaced by the HIP detector.  `detector` lets a caller (tests) plug in any detect_box-contract callable."""
    args = build_parser().parse_args(argv)
    if not args.target_list or not args.data_dir:
        print("--target_list and --data_dir are required", file=sys.stderr)
        return 2
    from .shard import init_from_env
    rank, world, local, dist = init_from_env(use_gpu=detector is None)
    own = None
    if detector is None:
        import torch
        torch.cuda.set_device(local if world > 1 and os.environ.get("GLOMSEG_ONE_GPU") != "1" else args.gpu_id)
        detector = own = load_detector(args.model, args.model_name, args.synthetic_weights)
    from .shard import abort_rank, finish_ranks
    try:
        split_all(args, detector, rank, world, dist)
    except BaseException:
        abort_rank(dist)      # a rank that fails alone must not wait for its peers in a barrier -- and leaves before any GPU teardown that could block (shard.abort_rank)
        if own is not None:
            own.close()
        raise
    if own is not None:
        own.close()
    finish_ranks(dist)
    return 0


if __name__ == '__main__':
    sys.exit(main())
