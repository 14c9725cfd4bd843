"""WSI mask compositor on the GPU: the consumer of the per-crop masks.

Restates what module/espnet/test/eval_wsi_segmentation.py does with them -- paste every crop's class
map onto the slide grid with max-compositing (overlay, :243-316), reduce to 1/8 scale
(generate_whole_img, :229), colour + blend over the slide (:230-240) and accumulate a WSI-level
confusion matrix (IOUEval.py:19-21) -- as device kernels behind the C ABI.

Two definitions of the 1/8 map are offered:
  * default: pixel (X, Y) = class at level-0 pixel (8X, 8Y).  That is exactly what the reference's INTER_NEAREST of a
    full 2400-px window produces.
  * ``reference_windows=True``: the reference's own window walk (:372-393), bit for bit: in the partial windows at the
    right / bottom edge of a slide that is not a multiple of 2400 px the nearest-neighbour step is
    (window size) / int(window size / 8) rather than 8, and windows with ``ymax > slide_width`` (a typo at :386 for
    slide_height) are never written -- on a slide taller than wide the bottom rows stay empty, as they do there.
Crop JSON decoding (the reference stores the original RGB crop in imageData) is replaced by taking the
class maps directly.
"""
import ctypes
import glob
import json
import os
import sys
from argparse import ArgumentParser

import numpy as np
import torch

from . import _lib, imageops

MAGNIFICATION = 8
WINDOW = 2400          # eval_wsi_segmentation.py: self.window_size


def _sp(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _axis_lut(size, limit, window=WINDOW, ds=MAGNIFICATION):
    """level-0 position shown by every cell of one axis of the 1/ds map under the reference's window walk
    (:372-387 windows, :229 cv2.resize INTER_NEAREST: src = min(floor(dst * (1 / (dst_size / src_size))), src_size - 1)).
    `limit`: windows whose upper edge exceeds it are skipped (-1)."""
    lut = np.full(int(size / ds), -1, dtype=np.int32)
    for ind in range(size // window + 1):
        lo = ind * window
        hi = size if ind == size // window else (ind + 1) * window
        if hi > limit or hi <= lo:
            continue
        n = int((hi - lo) / ds)
        if n <= 0:
            continue
        inv = 1.0 / (n / float(hi - lo))
        src = np.minimum(np.floor(np.arange(n) * inv).astype(np.int64), hi - lo - 1)
        lut[lo // ds: lo // ds + n] = lo + src
    return lut


def reference_window_luts(slide_width, slide_height):
    """(sx, sy) tables for gs_wsi_paste_max_lut.  The x windows are skipped when xmax > slide_width (never true, :378)
    and the y windows when ymax > slide_WIDTH (:386, the reference's typo, kept)."""
    return _axis_lut(slide_width, slide_width), _axis_lut(slide_height, slide_width)


class SlideCompositor:
    def __init__(self, slide_width, slide_height, device, ds=MAGNIFICATION, reference_windows=False):
        self.lib = _lib.load()
        self.ds = ds
        self.device = torch.device(device)
        self.map = torch.zeros((int(slide_height / ds), int(slide_width / ds)), dtype=torch.uint8, device=self.device)   # :371
        self.palette = torch.from_numpy(np.ascontiguousarray(imageops.PALETTE)).to(self.device)
        self.luts = None
        if reference_windows:
            if ds != MAGNIFICATION:
                raise ValueError("the reference's window walk is defined for its 1/8 map only")
            sx, sy = reference_window_luts(slide_width, slide_height)
            self.luts = (torch.from_numpy(sx).to(self.device), torch.from_numpy(sy).to(self.device))

    def paste_target(self):
        """the map as the _lib.PasteTarget the batched crop entries take (gs_espnet_segment_crops*): they paste a whole batch
        of crops per launch, with the same result as paste() per crop"""
        from .engine import paste_target
        return paste_target(self.map, self.ds, self.luts)

    def paste(self, crop_mask, x1, y1):
        """crop_mask: uint8 [h,w] (level-0 resolution) on the GPU or host; (x1,y1) level-0 origin of the box."""
        if not isinstance(crop_mask, torch.Tensor):
            crop_mask = torch.from_numpy(np.ascontiguousarray(crop_mask))
        crop_mask = crop_mask.to(self.device).contiguous()
        h, w = crop_mask.shape
        with torch.cuda.device(self.device):
            if self.luts is not None:
                _lib.check(self.lib.gs_wsi_paste_max_lut(self.map.data_ptr(), self.map.shape[0], self.map.shape[1], self.ds,
                                                         self.luts[0].data_ptr(), self.luts[1].data_ptr(),
                                                         crop_mask.data_ptr(), h, w, int(x1), int(y1), _sp(self.device)))
            else:
                _lib.check(self.lib.gs_wsi_paste_max(self.map.data_ptr(), self.map.shape[0], self.map.shape[1], self.ds,
                                                     crop_mask.data_ptr(), h, w, int(x1), int(y1), _sp(self.device)))

    def overlay(self, slide_bgr_small, wa=0.4, wb=0.6):
        """slide_bgr_small: uint8 [map_h,map_w,3] BGR (the 1/8-scale slide) -> blended prediction image."""
        img = slide_bgr_small if isinstance(slide_bgr_small, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(slide_bgr_small))
        img = img.to(self.device).contiguous()
        out = torch.empty_like(img)
        h, w = self.map.shape
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_overlay_classmap(img.data_ptr(), self.map.data_ptr(), h, w, self.palette.data_ptr(),
                                                    self.palette.shape[0], ctypes.c_float(wa), ctypes.c_float(wb),
                                                    out.data_ptr(), _sp(self.device)))
        return out

    def confusion(self, gt_map, classes=5, hist=None):
        """accumulate fast_hist(gt, pred) over the slide map; returns int64 [classes,classes] tensor."""
        gt = gt_map if isinstance(gt_map, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(gt_map))
        gt = gt.to(self.device).contiguous()
        if hist is None:
            hist = torch.zeros((classes, classes), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_confusion_u8(self.map.data_ptr(), gt.data_ptr(), self.map.numel(), classes, hist.data_ptr(),
                                                _sp(self.device)))
        return hist


# --------------------------------------------------------------------------- command line (prediction WSI)
def relabel(img):
    """eval_wsi_segmentation.py:49-55: Cityscapes ids back to 0..4 (identity on a map that is already 0..4)."""
    lut = np.arange(256, dtype=np.uint8)
    for src, dst in ((13, 4), (12, 3), (11, 2), (8, 1), (7, 0)):
        lut[src] = dst
    return lut[img]


def load_class_map(json_path):
    """The class map a segmentation JSON stands for: the PNG named by classMapPath (what segment.py always writes beside
    the JSON), else the decoded imageData -- the reference decodes imageData (:287-297), which only holds a class map
    when the segmentation was written with the variant commented out at VisualizeResults_iou.py:178 (--imageData classmap)."""
    import base64
    import io
    from PIL import Image
    with open(json_path) as f:
        data = json.load(f)
    if data.get("classMapPath"):
        return np.asarray(Image.open(os.path.join(os.path.dirname(json_path), data["classMapPath"])))
    if data.get("imageData"):
        img = np.asarray(Image.open(io.BytesIO(base64.b64decode(data["imageData"]))))
        if img.ndim == 2:
            return img
    raise ValueError("%s holds no class map (imageData is the original crop): re-run the segmentation with --imageData classmap, "
                     "or keep the *_classmap.png files beside the JSON" % json_path)


def build_parser():
    """argparse surface of eval_wsi_segmentation.py:397-422, every flag with the reference's dest and default.  The rebuilt
    path is the prediction branch (:430-431, generate_pred_wsi).  There the reference itself never reads --iou_threshold,
    --output_file, --start or --end (they are used by scan_files, the ground-truth branch, :108-118): they are accepted so
    that the reference's command lines (README.md:268-281, example/README.md:108-133) run unchanged, and have no effect.  The
    three ground-truth directories select the evaluation branch (:427-434), which needs the GT tooling that is out of scope
    (annotation XML, labelme shapes): accepted by the parser, refused by main() with an explanation."""
    p = ArgumentParser(description='merge cropped glomerular segmented images')
    p.add_argument('--staining', dest='staining', type=str, required=True)
    p.add_argument('--merged_detection_result_csv', dest='input_csv', type=str, required=True)
    p.add_argument('--target_list', dest='target_list', type=str, required=True)
    p.add_argument('--wsi_dir', dest='wsi_dir', type=str, required=True)
    p.add_argument('--segmentation_pred_json_dir', dest='seg_pred_json_dir', type=str, required=True)
    p.add_argument('--object_detection_gt_xml_dir', dest='ob_gt_xml_dir', type=str, default=None)
    p.add_argument('--segmentation_gt_json_dir', dest='seg_gt_json_dir', type=str, default=None)
    p.add_argument('--segmentation_gt_png_dir', dest='gt_png_dir', type=str, default=None)
    p.add_argument('--iou_threshold', dest='iou_threshold', type=float, default=0.01)
    p.add_argument('--output_file', dest='output_file', type=str, default='./output/seg_data_pred/seg_data_output.tsv')
    p.add_argument('--start', dest='start', type=int, default=0)
    p.add_argument('--end', dest='end', type=int, default=0)
    p.add_argument('--output_dir', dest='output_dir', type=str, default='./output/seg_data_pred')
    p.add_argument('--window_size', dest='window_size', type=int, default=2400)
    p.add_argument('--classes', dest='classes', type=int, default=5)
    p.add_argument('--no_save', dest='no_save', action='store_true')
    p.add_argument('--gpu_id', dest='gpu_id', type=int, default=0)
    return p


def slide_size(wsi_dir, file_key, target_meta):
    """slide.dimensions (:340-357) through OpenSlide when it is installed; else the size the target list records for the
    specimen (the PNG-branch metadata), else the size of a PNG slide found under wsi_dir/file_key times its downsample."""
    ndpi = glob.glob(os.path.join(wsi_dir, file_key, "*ndpi"))
    if ndpi:
        try:
            import openslide
            with openslide.open_slide(ndpi[0]) as s:
                return s.dimensions
        except ImportError:
            pass
    m = target_meta.get(file_key)
    if m and m["width"] > 0:
        return m["width"], m["height"]
    raise RuntimeError("size of slide %s unknown: OpenSlide is not installed and the target list has no metadata line for it" % file_key)


def generate_pred_wsi(args, out=sys.stdout):
    """generate_pred_wsi (:359-394): for every slide of the merged list, the crops' class maps max-composited under the
    reference's 2400-px window walk into the 1/8 map, coloured and blended over the 1/8 slide image -> <slide>_pred.jpg
    (+ <slide>_pred_classmap.png, additive)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from . import detect, merge
    if args.window_size != WINDOW:
        raise ValueError("the window walk is built for the reference's window size %d" % WINDOW)
    boxes_of, order = merge.read_merged_csv(args.input_csv)
    target_meta = {}
    if os.path.isfile(args.target_list):
        for line in open(args.target_list):
            m = detect.parse_target_line(line)
            if m:
                target_meta[m["specimen_id"].replace(' ', '')] = m
    os.makedirs(args.output_dir, exist_ok=True)
    dev = torch.device("cuda", args.gpu_id)
    results = {}
    for key in order:
        jsons = glob.glob(os.path.join(args.seg_pred_json_dir, key, "*.json"))
        w, h = slide_size(args.wsi_dir, key, target_meta)
        comp = SlideCompositor(w, h, dev, reference_windows=True)
        used = 0
        todo = []
        for b in boxes_of[key]:
            name = merge.crop_name(b)                                         # :268
            hits = [j for j in jsons if name in os.path.basename(j)]
            assert len(hits) <= 1
            if hits:                                                          # :272-274: a crop without a segmentation is skipped
                todo.append((b, hits[0]))
        # the class maps are decoded (JSON + PNG) by a few threads, then pasted in list order: max-compositing is order-free,
        # the asserts are not
        with ThreadPoolExecutor(max_workers=max(1, min(8, len(todo)))) as pool:
            maps = list(pool.map(lambda t: relabel(np.ascontiguousarray(load_class_map(t[1]), dtype=np.uint8)), todo))
        for (b, path), cm in zip(todo, maps):
            if cm.shape != (b[3] - b[1], b[2] - b[0]):
                raise ValueError("%s: class map %s does not fit its box %s" % (path, cm.shape, b[:4]))
            assert int(cm.max()) < args.classes                               # :314
            comp.paste(cm, b[0], b[1])
            used += 1
        mh, mw = comp.map.shape
        small = np.zeros((mh, mw, 3), dtype=np.uint8)
        pngs = glob.glob(os.path.join(args.wsi_dir, key, "*.PNG")) + glob.glob(os.path.join(args.wsi_dir, key, "*.png"))
        if pngs:          # the 1/8 slide image of the PNG branch, cropped / padded to the map
            with Image.open(pngs[0]) as im:
                rgb = np.asarray(im.convert("RGB"))
            hh, ww = min(mh, rgb.shape[0]), min(mw, rgb.shape[1])
            small[:hh, :ww] = rgb[:hh, :ww, ::-1]
        blended = comp.overlay(small).cpu().numpy()
        results[key] = {"map": comp.map.cpu().numpy(), "crops": used}
        if not args.no_save:
            Image.fromarray(np.ascontiguousarray(blended[:, :, ::-1])).save(os.path.join(args.output_dir, key + "_pred.jpg"),
                                                                            quality=95)     # cv2.imwrite's default JPEG quality (:394)
            Image.fromarray(results[key]["map"]).save(os.path.join(args.output_dir, key + "_pred_classmap.png"))
        print("{}: {} crops composited on a {} x {} map".format(key, used, mw, mh), file=out)
    return results


def main(argv=None):
    args = build_parser().parse_args(argv)
    # the reference takes the evaluation branch only when ALL three ground-truth directories are given (:427); with any of
    # them missing it composes the prediction WSI and ignores the rest
    if args.seg_gt_json_dir is not None and args.gt_png_dir is not None and args.ob_gt_xml_dir is not None:
        print("the ground-truth evaluation branch (scan_files: annotation XML, labelme shapes, the TSV of --output_file) is outside "
              "the rebuilt path; leave the three *_gt_* directories unset to compose the prediction WSI", file=sys.stderr)
        return 2
    generate_pred_wsi(args)
    return 0


if __name__ == '__main__':
    sys.exit(main())
