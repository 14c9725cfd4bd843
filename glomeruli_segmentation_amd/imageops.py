"""Host image helpers the per-patch driver needs around the GPU pass.  cv2 is not installed in
this image, so the two ``cv2.resize`` calls of the reference loop
(module/espnet/test/VisualizeResults_iou.py:114 INTER_LINEAR on float32, :129 INTER_NEAREST) are
restated here from OpenCV's documented sampling rules; both are identities when the crop already
has the network size (every BASELINE config).  Parity for non-identity resizes is unpinned (no
executable cv2 here) and says so in DESIGN.md.
"""
import numpy as np
from PIL import Image

# the reference's colour table, RGB (VisualizeResults_iou.py:20-44)
PALETTE = np.array([[0, 0, 0], [255, 0, 0], [0, 184, 0], [255, 255, 0], [0, 0, 255], [128, 64, 128], [244, 35, 232],
                    [70, 70, 70], [102, 102, 156], [190, 153, 153], [153, 153, 153], [250, 170, 30], [220, 220, 0],
                    [107, 142, 35], [152, 251, 152], [70, 130, 180], [220, 20, 60], [255, 0, 0], [0, 0, 142],
                    [0, 0, 70], [0, 60, 100], [0, 80, 100], [0, 0, 230], [119, 11, 32], [0, 0, 0]], dtype=np.uint8)


def imread_bgr(path):
    """cv2.imread equivalent for 8-bit images: HxWx3 uint8, BGR, alpha dropped (:103)."""
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def imwrite_bgr(path, bgr):
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)


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
    """cv2.resize(img, (width, height)) for a float32 HxWxC image, INTER_LINEAR."""
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
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src = min(floor(dst * src/dst), src-1)."""
    h, w = img.shape[:2]
    if (w, h) == (width, height):
        return img.copy()
    xs = np.minimum(np.floor(np.arange(width) * (w / float(width))).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(height) * (h / float(height))).astype(np.int64), h - 1)
    return img[ys][:, xs]


def normalise_then_resize(bgr_u8, mean, std, width, height):
    """VisualizeResults_iou.py:107-116 for a crop that is NOT already network-sized: the reference
    normalises at crop resolution, resizes the float image, then divides by 255.  Returns fp32
    CHW ready for the GS_IN_F32_NCHW entry."""
    img = bgr_u8.astype(np.float32)
    img -= np.asarray(mean, dtype=np.float32)
    img /= np.asarray(std, dtype=np.float32)
    img = resize_linear_f32(img, width, height)
    img /= 255
    return np.ascontiguousarray(img.transpose(2, 0, 1))


def colourise(class_map):
    """class map -> BGR colour image with the reference palette (:140-143)."""
    rgb = PALETTE[np.minimum(class_map, len(PALETTE) - 1)]
    return np.ascontiguousarray(rgb[:, :, ::-1])


def add_weighted(a, wa, b, wb):
    """cv2.addWeighted(a, wa, b, wb, 0) for uint8 images: saturate_cast<uchar>(round(...))."""
    v = a.astype(np.float32) * wa + b.astype(np.float32) * wb
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def relabel_city(img):
    """0..4 -> Cityscapes ids 7,8,11,12,13 (VisualizeResults_iou.py:54-81 restricted to 5 classes;
    the table is applied as one lookup, which equals the reference's in-place cascade for ids 0..19)."""
    lut = np.arange(256, dtype=np.uint8)
    src = [19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0]
    dst = [255, 33, 32, 31, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 17, 13, 12, 11, 8, 7]
    cur = np.arange(256, dtype=np.int64)
    for s, d in zip(src, dst):          # replay the cascade on the table, in the reference's order
        cur[cur == s] = d
    cur[cur == 255] = 0
    lut[:] = cur.astype(np.uint8)
    return lut[img]
