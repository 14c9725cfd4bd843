"""Where the time of gs_espnet_segment_crops_host goes (exploration): output allocation, staging, outputs on/off."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glomeruli_segmentation_amd.engine import EspnetEngine  # noqa: E402
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
eng = EspnetEngine({k: z[k] for k in z.files}, lanes=2)
mean, std = FOLD_MEAN_STD[1]
ex = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
base = [synth_tile(5000 + k, int(b[3] - b[1]), int(b[2] - b[0]), blobs=4) for k, b in enumerate(ex)]
crops = base * 16
pinned = [torch.from_numpy(c).pin_memory() for c in base] * 16
eng.segment_crops(crops[:96], mean, std)


def t(label, f, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    print("%-50s %7.1f ms  (%6.0f crops/s)" % (label, best * 1e3, len(crops) / best))


total = sum(c.shape[0] * c.shape[1] for c in crops)
t("torch.empty pinned %d MB" % (total >> 20), lambda: torch.empty(total, dtype=torch.uint8, pin_memory=True))
t("np.empty + touch", lambda: np.empty(total, dtype=np.uint8).fill(0))
t("pageable in, masks+hist", lambda: eng.segment_crops(crops, mean, std))
t("pageable in, hist only", lambda: eng.segment_crops(crops, mean, std, want_masks=False))
t("pinned in, masks+hist", lambda: eng.segment_crops(pinned, mean, std))
t("pinned in, hist only", lambda: eng.segment_crops(pinned, mean, std, want_masks=False))
t("pinned in, batch 64", lambda: eng.segment_crops(pinned, mean, std, batch=64, want_masks=False))
tiles = torch.from_numpy(np.stack([synth_tile(k) for k in range(32)] * 4)).pin_memory()
t("segment_host 128 network-size tiles (x3.5 = 448)", lambda: eng.segment_host(tiles, mean, std))
