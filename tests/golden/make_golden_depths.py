#!/usr/bin/env python3
"""Golden logits of the REFERENCE network at depths other than the shipped (p, q) = (2, 8).

Runs only in the build container (imports /root/reference/module/espnet/test/Model.py): for each (p, q) it builds the
reference's ESPNet(5, p, q), loads the seeded random state_dict of tests/conftest.random_state_dict (strict: every key of the
reference module must be there), and records the logits on one seeded noise tile.  Pins the oracle's -- and the HIP path's --
graph composition for depths other than (2, 8).  p = 0 or q = 0 cannot be pinned: the reference's own forward raises
UnboundLocalError there (Model.py:351-357 / :361-366 never assign output1 / output2 when the block list is empty); this
build's reading of those depths (the down-sampler's output stands in) is an extension, checked against its oracle only.

    python tests/golden/make_golden_depths.py        ->  tests/golden/depths.npz (arrays only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GS_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REF, "module", "espnet", "test"))

import Model as RefModel  # noqa: E402  (the reference's Model.py)
from conftest import random_state_dict  # noqa: E402
from glomeruli_segmentation_amd.synth import noise_tile  # noqa: E402

DEPTHS = [(2, 3), (1, 2), (3, 1), (1, 1)]
MEAN, STD = (120.0, 130.0, 110.0), (60.0, 55.0, 70.0)

torch.set_grad_enabled(False)
out = {"depths": np.array(DEPTHS), "mean": np.array(MEAN, np.float32), "std": np.array(STD, np.float32)}
tile = noise_tile(77, 48, 104)
out["tile"] = tile
img = tile.astype(np.float32)
for j in range(3):                       # VisualizeResults_iou.py:107-117
    img[:, :, j] -= MEAN[j]
for j in range(3):
    img[:, :, j] /= STD[j]
img /= 255
x = torch.from_numpy(np.ascontiguousarray(img.transpose((2, 0, 1)))).unsqueeze(0)
for p, q in DEPTHS:
    sd = random_state_dict(p, q, seed=10 * p + q)
    net = RefModel.ESPNet(5, p, q)
    msg = net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    assert not msg.missing_keys and not msg.unexpected_keys, msg
    net.eval()
    out["logits_p%d_q%d" % (p, q)] = net(x)[0].numpy()
np.savez_compressed(os.path.join(HERE, "depths.npz"), **out)
print("depths.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "depths.npz")) / 1024.0))
