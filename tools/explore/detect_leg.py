"""Where the detect leg of tools/bench_slide.py goes (exploration): host pipeline alone, per batch size, and the row arithmetic."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glomeruli_segmentation_amd import detect  # noqa: E402
from glomeruli_segmentation_amd.detector import FrcnnDetector, synthetic_weights  # noqa: E402

det = FrcnnDetector(synthetic_weights(0))
rng = np.random.default_rng(0)
wins = [rng.integers(0, 256, (1098, 1098, 3), dtype=np.uint8) for _ in range(36)]
pinned = [torch.from_numpy(w).pin_memory() for w in wins]


def t(label, f, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        r = f()
        best = min(best, time.perf_counter() - t0)
    print("%-44s %7.2f ms  (%6.0f windows/s)" % (label, best * 1e3, 36 / best))
    return r


for b in (16, 12, 18, 36):
    t("detect_host pageable batch %d" % b, lambda: det.detect_host(wins, batch=b))
t("detect_host pinned batch 16", lambda: det.detect_host(pinned, batch=16))
t("detect_host pinned batch 12", lambda: det.detect_host(pinned, batch=12))
dev = torch.from_numpy(np.stack(wins[:16])).cuda()
torch.cuda.synchronize()
t("forward_device 16 resident (x2.25 = 36)", lambda: (det.forward_device(dev), torch.cuda.synchronize()))
boxes, scores, classes, num = det.detect_host(wins, batch=16)
plan = detect.plan_windows(40000, 40000, 0.2277, 0.2277, 8.0, 2000, 0.1)


def rows():
    out = []
    for k, (i, j, xs, ys) in enumerate(plan.origins()):
        bs = detect.boxes_from_detector(boxes[k], scores[k], plan.window_x, plan.window_y, 0.2)
        out.extend(detect.csv_rows(bs, xs, ys, plan.downsample, "site", "slide", "slide.ndpi"))
    return out


r = t("threshold + denormalise + CSV rows (python)", rows)
print(len(r), "rows")
