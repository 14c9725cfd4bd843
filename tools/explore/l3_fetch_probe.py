#!/usr/bin/env python3
"""A few resident forwards at batch B (argv[1]) for counter passes: how does the beyond-L2 fetch volume of the level-3
branch kernels depend on the number of images an XCD has in flight?  (tools/explore/l3_fetch_probe.sh)"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import torch
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
eng = EspnetEngine({k: z[k] for k in z.files})
mean, std = FOLD_MEAN_STD[1]
tiles = torch.from_numpy(np.stack([synth_tile(i % 8) for i in range(B)])).cuda()
for _ in range(4):
    eng.segment(tiles, mean, std)
torch.cuda.synchronize()
