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
