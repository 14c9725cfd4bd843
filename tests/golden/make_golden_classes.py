#!/usr/bin/env python3
"""Golden logits of the REFERENCE network at class counts other than the shipped 5.

Runs only in the build container (imports /root/reference/module/espnet/test/Model.py).  The reference's operator takes
`classes` freely -- `ESPNet(classes=20, p=2, q=3)` is the constructor's DEFAULT (Model.py:311; encoder :246) and the driver
passes --classes through (VisualizeResults_iou.py:274,315).  For each case it builds the reference module -- the first one
literally as `ESPNet()` with no arguments -- loads the seeded random state_dict of tests/conftest.random_state_dict (strict:
every key of the reference module must be there) and records logits and the first-max class map (VisualizeResults_iou.py:128)
on one seeded noise tile; and ESPNet-C as `ESPNet_Encoder()` = (20, 5, 3).  Pins the oracle's and the HIP path's decoder for
2 <= classes <= 20.

    python tests/golden/make_golden_classes.py        ->  tests/golden/classes.npz (arrays only)
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

# (classes, p, q, tile height, tile width); the first row is the reference constructor's default arguments
CASES = [(20, 2, 3, 48, 104), (7, 2, 3, 48, 104), (2, 1, 1, 32, 72), (12, 2, 2, 32, 72), (16, 1, 2, 32, 72), (3, 2, 8, 32, 72)]
MEAN, STD = (120.0, 130.0, 110.0), (60.0, 55.0, 70.0)

torch.set_grad_enabled(False)
out = {"cases": np.array(CASES), "mean": np.array(MEAN, np.float32), "std": np.array(STD, np.float32)}
for k, (classes, p, q, th, tw) in enumerate(CASES):
    tile = noise_tile(300 + k, th, tw)
    img = tile.astype(np.float32)
    for j in range(3):                       # VisualizeResults_iou.py:107-117
        img[:, :, j] -= MEAN[j]
    for j in range(3):
        img[:, :, j] /= STD[j]
    img /= 255
    x = torch.from_numpy(np.ascontiguousarray(img.transpose((2, 0, 1)))).unsqueeze(0)
    sd = random_state_dict(p, q, classes=classes, seed=1000 + classes)
    net = RefModel.ESPNet() if k == 0 else RefModel.ESPNet(classes, p, q)
    msg = net.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in sd.items()})
    assert not msg.missing_keys and not msg.unexpected_keys, msg
    net.eval()
    lg = net(x)
    assert tuple(lg.shape) == (1, classes, th, tw)
    tag = "c%d" % classes
    out["tile_" + tag] = tile
    out["logits_" + tag] = lg[0].numpy()
    out["mask_" + tag] = lg[0].max(0)[1].byte().numpy()          # :128
# ESPNet-C with the constructor's own defaults, ESPNet_Encoder() = (classes 20, p 5, q 3) (Model.py:246): logits at 1/8 scale
tile = noise_tile(350, 48, 104)
img = tile.astype(np.float32)
for j in range(3):
    img[:, :, j] -= MEAN[j]
for j in range(3):
    img[:, :, j] /= STD[j]
img /= 255
x = torch.from_numpy(np.ascontiguousarray(img.transpose((2, 0, 1)))).unsqueeze(0)
sd = {n[len("encoder."):]: v for n, v in random_state_dict(5, 3, classes=20, seed=2020).items() if n.startswith("encoder.")}
enc = RefModel.ESPNet_Encoder()
msg = enc.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in sd.items()})
assert not msg.missing_keys and not msg.unexpected_keys, msg
enc.eval()
out["tile_enc"] = tile
out["logits_enc"] = enc(x)[0].numpy()
assert out["logits_enc"].shape == (20, 6, 13)
np.savez_compressed(os.path.join(HERE, "classes.npz"), **out)
print("classes.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "classes.npz")) / 1024.0))
