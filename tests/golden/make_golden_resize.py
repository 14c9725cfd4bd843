"""Writes tests/golden/resize.npz: cv2.resize-shaped known answers from an implementation this repo did not write.

cv2 is not installed here; torch is.  torch.nn.functional.interpolate(mode="bilinear", align_corners=False,
antialias=False) samples with the same half-pixel rule as cv2.resize(INTER_LINEAR) on float input, and mode="nearest"
with cv2's INTER_NEAREST rule src = floor(dst * src_size / dst_size).  Where they can differ: torch evaluates the four
bilinear products in one expression and cv2 in two passes (a rounding-order difference of a few 1e-7 relative), and
torch computes the nearest scale in float32: for the size pairs below it picks the same source pixels as the float64
rule OpenCV evaluates (asserted when the file is written).  The one disagreement met while choosing sizes: 256 -> 194
columns, destination 97 (97 * 256 / 194 is exactly 128; the float32 scale yields 127) -- such a pair would pin torch's
rounding, not cv2's rule, and is not used.

    python tests/golden/make_golden_resize.py

Inputs and outputs are arrays only (no reference source): a crop, its normalised + resized tensor in the reference's
order (VisualizeResults_iou.py:107-116: (x - mean) / std at crop size, resize, / 255), and a class map resized back.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
MEAN = np.array([204.60071, 170.19359, 199.57469], dtype=np.float32)
STD = np.array([20.61257, 42.92207, 28.401505], dtype=np.float32)
CASES = [(75, 105, 64, 128), (192, 384, 64, 128), (37, 53, 64, 128), (128, 256, 64, 128), (97, 515, 32, 64),
         (277, 200, 128, 256)]   # (crop h, crop w, net h, net w): up, x3 down, ragged up, x2 down, mixed, up / down per axis


def main():
    rng = np.random.default_rng(2024)
    out = {"mean": MEAN, "std": STD, "cases": np.array(CASES, dtype=np.int32)}
    for k, (h, w, oh, ow) in enumerate(CASES):
        crop = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        x = (torch.from_numpy(crop.astype(np.float32)) - torch.from_numpy(MEAN)) / torch.from_numpy(STD)
        y = F.interpolate(x.permute(2, 0, 1)[None], size=(oh, ow), mode="bilinear", align_corners=False, antialias=False)[0]
        y = y / 255
        cmap = rng.integers(0, 5, size=(oh, ow), dtype=np.uint8)
        back = F.interpolate(torch.from_numpy(cmap)[None, None].float(), size=(h, w), mode="nearest")[0, 0].to(torch.uint8)
        # the float64 form of the same rule (what OpenCV evaluates): must pick the same source pixels
        ys = np.minimum(np.floor(np.arange(h) * (1.0 / (h / float(oh)))).astype(np.int64), oh - 1)
        xs = np.minimum(np.floor(np.arange(w) * (1.0 / (w / float(ow)))).astype(np.int64), ow - 1)
        assert np.array_equal(back.numpy(), cmap[ys][:, xs]), "float32 / float64 nearest rules differ for case %d" % k
        out["crop_%d" % k] = crop
        out["net_%d" % k] = y.numpy().astype(np.float32)
        out["cmap_%d" % k] = cmap
        out["back_%d" % k] = back.numpy()
    np.savez_compressed(os.path.join(HERE, "resize.npz"), **out)
    print("resize.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "resize.npz")) / 1024.0))


if __name__ == "__main__":
    main()
