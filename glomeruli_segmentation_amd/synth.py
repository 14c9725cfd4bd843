"""Seeded synthetic stained-tissue tiles (no dataset can be downloaded here).

The generator is the one SURVEY.md section 8(d) specifies: a PAS-like base colour
plus a few Gaussian blobs plus pixel noise.  It produced three of the five
classes and realistic logit margins with the shipped fold weights.  Tiles are
uint8 BGR, HxWx3, exactly what ``cv2.imread`` hands the reference's per-patch
loop (reference: module/espnet/test/VisualizeResults_iou.py:103).
"""
import hashlib

import numpy as np

BASE_BGR = (204.0, 170.0, 199.0)
BLOB_SCALE = (20.0, 43.0, 28.0)

# per-fold BGR mean / std of the training images (reference: README.md:243-249)
FOLD_MEAN_STD = {
    1: ((204.60071, 170.19359, 199.57469), (20.61257, 42.92207, 28.401505)),
    2: ((202.38148, 167.13171, 198.10599), (20.704079, 42.958416, 28.366297)),
    3: ((203.12099, 167.813, 198.50894), (21.038654, 43.769535, 29.034416)),
    4: ((203.66399, 167.94217, 198.58081), (20.96783, 43.556736, 28.838718)),
    5: ((204.49896, 169.03307, 199.22058), (20.547842, 42.86628, 27.966227)),
}


def synth_tile(seed, height=512, width=1024, blobs=12):
    """One uint8 BGR tile, deterministic in (seed, height, width, blobs)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.empty((height, width, 3), dtype=np.float32)
    img[:] = np.asarray(BASE_BGR, dtype=np.float32)
    for _ in range(blobs):
        cy = rng.uniform(0, height)
        cx = rng.uniform(0, width)
        sigma = rng.uniform(20.0, 160.0) * (min(height, width) / 512.0)
        colour = rng.standard_normal(3) * np.asarray(BLOB_SCALE) * 1.5
        g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2.0 * sigma * sigma))
        img += g[:, :, None] * colour.astype(np.float32)[None, None, :]
    img += rng.normal(0.0, 6.0, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def noise_tile(seed, height=512, width=1024):
    """Pure uint8 noise tile (stress input)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8)


def synth_batch(seeds, height=512, width=1024):
    return np.stack([synth_tile(s, height, width) for s in seeds], axis=0)


def tile_digest(tile):
    return hashlib.sha256(np.ascontiguousarray(tile).tobytes()).hexdigest()
