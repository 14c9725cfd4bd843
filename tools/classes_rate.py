#!/usr/bin/env python3
"""Resident throughput of the forward at class counts other than five (the two-kernel decoder tail, csrc/espnet.hip forward_impl):
ESPNet(classes, 2, 8) with random weights (tests/conftest.random_state_dict), 32 uint8 tiles of 1024x512 -> masks + counts, one
lane; five classes (the fused tail) in the same run for comparison.      python tools/classes_rate.py [--out profiles/r05_classes_rate.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import random_state_dict  # noqa: E402
from glomeruli_segmentation_amd.engine import EspnetEngine  # noqa: E402
from glomeruli_segmentation_amd.synth import synth_tile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--classes", default="5,2,4,7,8,12,16,20")
    a = ap.parse_args()
    tiles = torch.from_numpy(np.stack([synth_tile(k) for k in range(32)])).cuda()
    mean, std = (120.0, 130.0, 110.0), (60.0, 55.0, 70.0)
    rows = []
    for c in [int(v) for v in a.classes.split(",")]:
        eng = EspnetEngine(random_state_dict(2, 8, classes=c, seed=c), classes=c, p=2, q=8)
        mask = torch.empty((32, 512, 1024), dtype=torch.uint8, device="cuda")
        hist = torch.empty((32, c), dtype=torch.int64, device="cuda")
        for _ in range(3):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        eng.profile(True)
        for _ in range(10):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        kern = {k["name"]: round(k["total_ms"] / max(k["launches"], 1), 4) for k in eng.profile_read() if k["launches"] and "dec" in k["name"]}
        eng.profile(False)
        assert int(hist.sum()) == 32 * 512 * 1024
        rows.append({"classes": c, "ms_per_32_tiles": round(ms, 3), "patches_per_s": round(32e3 / ms, 1), "decoder_kernels_ms": kern})
        print(rows[-1], flush=True)
        eng.close()
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"what": "ESPNet(classes, 2, 8), random weights, 32 x 1024x512 uint8 tiles resident -> masks + counts, one lane",
                       "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
