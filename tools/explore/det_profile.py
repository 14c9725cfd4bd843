import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from glomeruli_segmentation_amd.detector import FrcnnDetector, synthetic_weights
from glomeruli_segmentation_amd.synth import synth_tile
torch.cuda.set_device(0)
det = FrcnnDetector(synthetic_weights(0))
wins = torch.from_numpy(np.stack([synth_tile(200 + i, 1000, 1000, blobs=8)[:, :, ::-1].copy() for i in range(4)] * 4)).cuda()
for _ in range(4):
    det.forward_device(wins)
torch.cuda.synchronize()
