"""TEST INFRASTRUCTURE -- CPU restatements of the cv2-shaped stages either side of the ESPNet forward.
Only tests/ may import this module; the product package has no CPU path for these stages (they run as HIP
kernels: gs_crop_preprocess, gs_mask_resize_nearest, gs_wsi_paste_max[_lut]).

cv2 is not installed in this image, so the two ``cv2.resize`` calls of the reference loop
(module/espnet/test/VisualizeResults_iou.py:114 INTER_LINEAR on float32, :129 INTER_NEAREST) are restated here from
OpenCV's sampling rules and PINNED against an independent implementation: tests/golden/resize.npz holds the outputs
of torch.nn.functional.interpolate (bilinear, align_corners=False, antialias=False = the same half-pixel rule;
nearest = floor(dst * scale)) on fixed inputs, written by tests/golden/make_golden_resize.py.

One case where OpenCV itself takes another route: for an exact 2x down-scale in both directions (e.g. a 2048x1024 crop to
1024x512) cv2.resize silently switches INTER_LINEAR to its INTER_AREA code (resize.cpp: `if (interpolation == INTER_LINEAR
&& is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA`).  The result is mathematically the same --
both are the mean of the 2x2 block, the bilinear taps being 0.5 / 0.5 there -- but the area code sums the four values and
multiplies by 0.25 where the linear code blends rows then columns, so the last bit can differ.  This restatement (and the GPU
kernels) use the linear form everywhere; with cv2 absent neither rounding order can be pinned.
"""
import numpy as np


def _linear_taps(dst, src):
    scale = src / float(dst)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)   # half-pixel centres, no antialias
    i0 = np.floor(f).astype(np.int64)
    w1 = f - i0.astype(np.float32)
    lo = i0 < 0
    i0[lo] = 0
    w1[lo] = 0.0
    hi = i0 >= src - 1
    i0[hi] = src - 1
    w1[hi] = 0.0
    i1 = np.minimum(i0 + 1, src - 1)
    return i0, i1, w1


def resize_linear_f32(img, width, height):
    """cv2.resize(img, (width, height)) for a float32 HxWxC image, INTER_LINEAR (VisualizeResults_iou.py:114)."""
    h, w = img.shape[:2]
    if (w, h) == (width, height):
        return img.copy()
    x0, x1, wx = _linear_taps(width, w)
    y0, y1, wy = _linear_taps(height, h)
    one = np.float32(1.0)
    rows = img[:, x0] * (one - wx)[None, :, None] + img[:, x1] * wx[None, :, None]   # horizontal pass first
    out = rows[y0] * (one - wy)[:, None, None] + rows[y1] * wy[:, None, None]
    return out.astype(np.float32)


def resize_nearest(img, width, height):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST) (:129): src = min(floor(dst * (1 / (dst_size / src_size))), src-1)."""
    h, w = img.shape[:2]
    if (w, h) == (width, height):
        return img.copy()
    xs = np.minimum(np.floor(np.arange(width) * (1.0 / (width / float(w)))).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(height) * (1.0 / (height / float(h)))).astype(np.int64), h - 1)
    return img[ys][:, xs]


def normalise_then_resize(bgr_u8, mean, std, width, height):
    """VisualizeResults_iou.py:107-116 for a crop that is NOT already network-sized: the reference
    normalises at crop resolution, resizes the float image, then divides by 255.  Returns fp32 CHW."""
    img = bgr_u8.astype(np.float32)
    img -= np.asarray(mean, dtype=np.float32)
    img /= np.asarray(std, dtype=np.float32)
    img = resize_linear_f32(img, width, height)
    img /= 255
    return np.ascontiguousarray(img.transpose(2, 0, 1))


def reference_wsi_pred_map(crops, boxes, slide_width, slide_height, window=2400, ds=8):
    """eval_wsi_segmentation.py generate_pred_wsi (:359-393) for class maps: walk the 2400-px windows (x outer, y inner;
    the last window of an axis is the partial rest), skip a window when `ymax > slide_width` (the reference's typo, :386),
    max-composite every crop that overlaps the window (overlay, :260-312, margin 0 for predictions), reduce the window by
    cv2.resize INTER_NEAREST to (int(w/8), int(h/8)) (:229) and write it at [ymin//8:ymax//8, xmin//8:xmax//8] (:236-240).
    crops: list of uint8 [h,w] class maps; boxes: list of (x1,y1,x2,y2) level-0 boxes.  Returns the uint8 1/8 map."""
    out = np.zeros((int(slide_height / ds), int(slide_width / ds)), dtype=np.uint8)
    for x_ind in range(slide_width // window + 1):
        xmin = x_ind * window
        xmax = slide_width if x_ind == slide_width // window else (x_ind + 1) * window
        if xmax > slide_width:
            continue
        for y_ind in range(slide_height // window + 1):
            ymin = y_ind * window
            ymax = slide_height if y_ind == slide_height // window else (y_ind + 1) * window
            if ymax > slide_width:          # sic
                continue
            if xmax <= xmin or ymax <= ymin or int((xmax - xmin) / ds) == 0 or int((ymax - ymin) / ds) == 0:
                continue                    # (an empty window: cv2.resize has nothing to produce)
            wnd = np.zeros((ymax - ymin, xmax - xmin), dtype=np.uint8)
            for m, (bx1, by1, bx2, by2) in zip(crops, boxes):
                ix1, iy1, ix2, iy2 = max(bx1, xmin), max(by1, ymin), min(bx2, xmax), min(by2, ymax)
                if ix2 <= ix1 or iy2 <= iy1:
                    continue
                sub = m[iy1 - by1:iy2 - by1, ix1 - bx1:ix2 - bx1]
                wnd[iy1 - ymin:iy2 - ymin, ix1 - xmin:ix2 - xmin] = np.maximum(wnd[iy1 - ymin:iy2 - ymin, ix1 - xmin:ix2 - xmin], sub)
            small = resize_nearest(wnd, int((xmax - xmin) / ds), int((ymax - ymin) / ds))
            out[ymin // ds:ymax // ds, xmin // ds:xmax // ds] = small
    return out
