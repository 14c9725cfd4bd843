#!/usr/bin/env python3
"""Experiment: the bench's two-lanes loop with PARTITIONED lanes (gs_espnet_partition_lanes: each lane on a CU-masked stream -- 16 of
the 32 CUs of every XCD -- with its launches sized for that half) against plain streams (the lanes take turns on the whole chip).
        python tools/explore/cumask_lanes.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from glomeruli_segmentation_amd.engine import EspnetEngine  # noqa: E402
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD  # noqa: E402


def masked_stream(hip, words, lo, hi):
    mask = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        mask[i] = 0xffffffff
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
    mean, std = FOLD_MEAN_STD[1]
    tiles = torch.from_numpy(bench.make_batches(0)).to(dev)
    eng = EspnetEngine(bench.load_weights(), lanes=2)
    eng.reserve(32, 512, 1024)
    masks = [torch.empty((32, 512, 1024), dtype=torch.uint8, device=dev) for _ in range(2)]
    hists = torch.zeros((23, 32, 5), dtype=torch.int64, device=dev)
    nonlocal_holder = None

    def step(i, lanes):
        if lanes == -1:      # every step on lane 0's stream
            eng.segment(tiles[i % 4], mean, std, out_mask=masks[0], out_hist=hists[i], lane=0)
            return
        eng.segment(tiles[i % 4], mean, std, out_mask=masks[i % lanes], out_hist=hists[i], lane=(i % lanes) if lanes > 1 else None)

    def run(tag, lanes, reps=7, steps=20):
        for i in range(3):
            step(i, lanes)
        eng.wait_lanes()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for i in range(steps):
                step(i, lanes)
            eng.wait_lanes()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps * 1e3)
        ts.sort()
        print("%-34s median %.3f ms/step  (min %.3f max %.3f)  %.0f patches/s" % (tag, ts[len(ts) // 2], ts[0], ts[-1], 32e3 / ts[len(ts) // 2]), flush=True)

    # (a CU-masked stream is a BLOCKING stream: anything recorded on the legacy default stream waits for it and it for that -- so the
    # loop runs with a non-blocking side stream as torch's current stream, or the lanes' wait_stream(current) would chain them)
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for rnd in range(3):
            eng.partition_lanes(1)
            run("plain streams, two lanes", 2)
            run("one lane", 1)
            eng.partition_lanes(2)
            run("partitioned lanes (2 x 128 CUs)", 2)
            run("lane 0 alone on its 128 CUs", -1)
        tiles = masks = hists = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        eng.partition_lanes(1)
    eng.close()


if __name__ == "__main__":
    main()
