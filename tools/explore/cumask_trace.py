#!/usr/bin/env python3
"""A few two-lane steps with PARTITIONED lanes, to be run under `rocprofv3 --kernel-trace`: do the kernels of the two lanes overlap in time?
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cumask_trace -o t -- python3 tools/explore/cumask_trace.py [parts]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from glomeruli_segmentation_amd.engine import EspnetEngine  # noqa: E402
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
mean, std = FOLD_MEAN_STD[1]
tiles = torch.from_numpy(bench.make_batches(0)).to(dev)
eng = EspnetEngine(bench.load_weights(), lanes=2)
eng.reserve(32, 512, 1024)
masks = [torch.empty((32, 512, 1024), dtype=torch.uint8, device=dev) for _ in range(2)]
hists = torch.zeros((16, 32, 5), dtype=torch.int64, device=dev)
eng.partition_lanes(parts)
for i in range(8):
    eng.segment(tiles[i % 4], mean, std, out_mask=masks[i % 2], out_hist=hists[i], lane=i % 2)
eng.wait_lanes()
torch.cuda.synchronize()
del tiles, masks, hists          # (tensors recorded on the lanes' streams go before the streams do)
torch.cuda.empty_cache()
eng.partition_lanes(1)
eng.close()
print("done", flush=True)
