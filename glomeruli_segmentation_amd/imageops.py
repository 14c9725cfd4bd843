"""Host image helpers the per-patch driver needs around the GPU pass: file I/O in cv2's channel order, the
reference's palette, addWeighted and the Cityscapes relabel.  The two ``cv2.resize`` calls of the reference loop
(module/espnet/test/VisualizeResults_iou.py:114, :129) run on the GPU (gs_crop_preprocess / gs_mask_resize_nearest);
their CPU restatement is test infrastructure and lives in oracle/image_oracle.py.
"""
import os

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
    """cv2.imwrite for 8-bit BGR images with cv2's default encoder parameters (the reference passes none, :147-150,231):
    JPEG quality 95 (IMWRITE_JPEG_QUALITY; PIL's own default would be 75) and PNG compression level 1
    (IMWRITE_PNG_COMPRESSION: "best speed"; PIL's default 6 takes three times as long for a few per cent of size).  The
    encoders differ, so files are not byte-identical to cv2's; the pixels of a PNG are, a JPEG's within its quantisation."""
    im = Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1]))
    ext = os.path.splitext(path)[1].lower()
    if ext in (".jpg", ".jpeg"):
        im.save(path, quality=95)
    elif ext == ".png":
        im.save(path, compress_level=1)
    else:
        im.save(path)


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
