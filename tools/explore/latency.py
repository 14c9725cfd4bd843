"""Single-tile and small-batch latency of the resident forward (exploration): is it launch-bound?"""
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
eng = EspnetEngine({k: z[k] for k in z.files})
mean, std = FOLD_MEAN_STD[1]
for n in (1, 2, 4, 8, 16, 32):
    tiles = torch.from_numpy(np.stack([synth_tile(k) for k in range(n)])).cuda()
    mask = torch.empty((n, 512, 1024), dtype=torch.uint8, device="cuda")
    hist = torch.empty((n, 5), dtype=torch.int64, device="cuda")
    for _ in range(5):
        eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
    t_enq = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps
    # one call at a time, synchronised: latency
    t0 = time.perf_counter()
    for _ in range(20):
        eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        torch.cuda.synchronize()
    t_lat = (time.perf_counter() - t0) / 20
    print("n=%2d  enqueue %.3f ms  throughput %.3f ms/call (%.0f tiles/s)  latency %.3f ms" % (n, t_enq * 1e3, t_all * 1e3, n / t_all, t_lat * 1e3))
