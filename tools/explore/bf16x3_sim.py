#!/usr/bin/env python3
"""Exploratory (never the headline): would a 3-term bf16 split of the convolutions keep parity?

fp32 MFMA runs at 1/16 of the bf16 matrix rate on gfx950.  A product a*b of two fp32 numbers can be approximated by
three bf16 MFMAs with fp32 accumulation: a = ah + al, b = bh + bl (ah = bf16(a), al = bf16(a - ah), ...) and
a*b ~ ah*bh + ah*bl + al*bh (the dropped al*bl term is ~2^-16 relative).  bf16 x bf16 products are exact in fp32, so
this script simulates exactly what such kernels would compute -- F.conv2d in fp32 on bf16-representable operands, three
convolutions summed -- through the torch port of the graph, on the golden tiles, and reports logits error and mask
agreement against the reference's goldens.  Modes:
    branches   only the dilated 3x3 branch convolutions of the level-2 / level-3 blocks (77 % of the FLOPs)
    all        every convolution / deconvolution of the network
    bf16x1     one bf16 pass (ah*bh) for comparison: what plain bf16 MFMA kernels would give

    python tools/explore/bf16x3_sim.py            (CPU, ~2 minutes)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD  # noqa: E402
from oracle import espnet_oracle as orc  # noqa: E402
from oracle import espnet_torch_port as port  # noqa: E402

_conv2d, _convT = F.conv2d, F.conv_transpose2d
MODE = {"terms": 3, "which": "branches"}


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def conv_split(fn, x, w, *args, **kw):
    xh, xl = split(x)
    wh, wl = split(w)
    y = fn(xh, wh, *args, **kw)
    if MODE["terms"] == 3:
        y = y + fn(xh, wl, *args, **kw) + fn(xl, wh, *args, **kw)
    return y


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    is_branch = w.shape[2] == 3 and w.shape[0] in (12, 16, 25, 28) and stride == 1
    if MODE["which"] == "all" or is_branch:
        return conv_split(_conv2d, x, w, b, stride, padding, dilation)
    return _conv2d(x, w, b, stride, padding, dilation)


def convT(x, w, b=None, stride=1):
    if MODE["which"] == "all":
        return conv_split(_convT, x, w, b, stride)
    return _convT(x, w, b, stride)


def main():
    torch.set_num_threads(8)
    F.conv2d, F.conv_transpose2d = conv2d, convT
    z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
    sd = {k: torch.from_numpy(z[k]) for k in z.files}
    small = np.load(os.path.join(REPO, "tests", "golden", "small_fold1.npz"))
    masks = np.load(os.path.join(REPO, "tests", "golden", "masks_fold1.npz"))
    mean, std = FOLD_MEAN_STD[1]
    from glomeruli_segmentation_amd.synth import synth_tile
    print("%-22s %14s %16s %14s %12s" % ("mode", "logits max err", "(tile b 128x256)", "mask agreement", "mIoU vs ref"))
    for which, terms in (("none", 3), ("branches", 3), ("all", 3), ("branches", 1), ("all", 1)):
        MODE["which"], MODE["terms"] = which, terms
        err = 0.0
        for tag in ("a", "b", "c"):
            lg = port.espnet_forward(port.preprocess(small["tile_" + tag][None], mean, std), sd)[0].numpy()
            err = max(err, float(np.abs(lg - small["logits_" + tag]).max()))
        conf = np.zeros((5, 5), dtype=np.int64)
        for seed in range(2):
            lg = port.espnet_forward(port.preprocess(synth_tile(seed)[None], mean, std), sd)[0].numpy()
            conf += orc.confusion(orc.argmax(lg), masks["mask_%d" % seed])
        name = "fp32 (torch port)" if which == "none" else "bf16x%d %s" % (terms, which)
        print("%-22s %14.3e %16s %14.7f %12.6f" % (name, err, "", np.trace(conf) / conf.sum(), orc.present_class_miou(conf)))


if __name__ == "__main__":
    main()
