#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container, where /root/reference is mounted: it imports
the reference's module/espnet/test/Model.py, loads models/espnet_fold{1..5}.pth
and records inputs/outputs.  Only arrays are written (data, never source); the
GPU box replays them without the reference.  Pre/post-processing restates
module/espnet/test/VisualizeResults_iou.py:107-119,128,151-156 in numpy because
that script imports cv2/labelme, which are not installed (at 1024x512 the
cv2.resize calls are identities).

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GS_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, "module", "espnet", "test"))

import Model as RefModel  # noqa: E402  (the reference's Model.py)

from glomeruli_segmentation_amd.synth import (FOLD_MEAN_STD, synth_tile,  # noqa: E402
                                               tile_digest)

torch.set_grad_enabled(False)
torch.manual_seed(0)


def load_fold(fold):
    sd = torch.load(os.path.join(REF, "models", "espnet_fold%d.pth" % fold), map_location="cpu")
    net = RefModel.ESPNet(5, 2, 8)
    msg = net.load_state_dict(sd)
    assert not msg.missing_keys and not msg.unexpected_keys
    net.eval()
    return net, sd


def preprocess(tile_u8, mean, std):
    """VisualizeResults_iou.py:107-119 (BGR order kept, no channel swap)."""
    img = tile_u8.astype(np.float32)
    for j in range(3):
        img[:, :, j] -= mean[j]
    for j in range(3):
        img[:, :, j] /= std[j]
    img /= 255
    img = img.transpose((2, 0, 1))
    return torch.from_numpy(np.ascontiguousarray(img)).unsqueeze(0)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


def main():
    nets = {}
    for fold in range(1, 6):
        net, sd = load_fold(fold)
        nets[fold] = net
        arrays = {k: v.numpy() for k, v in sd.items()}
        save("weights_fold%d.npz" % fold, **arrays)
        h = hashlib.sha256()
        for k, v in sd.items():
            if v.dtype == torch.float32:
                h.update(k.encode())
                h.update(np.ascontiguousarray(v.numpy()).tobytes())
        mean, std = FOLD_MEAN_STD[fold]
        out = {"weights_sha256": np.frombuffer(bytes.fromhex(h.hexdigest()), dtype=np.uint8)}
        # (i) full-size masks + (iv) per-class pixel counts, seeds 0..3
        for seed in range(4):
            tile = synth_tile(seed)
            logits = net(preprocess(tile, mean, std))
            mask = logits[0].max(0)[1].byte().numpy()      # VisualizeResults_iou.py:128
            out["mask_%d" % seed] = mask
            out["hist_%d" % seed] = np.bincount(mask.ravel(), minlength=5).astype(np.int64)
            out["digest_%d" % seed] = np.frombuffer(bytes.fromhex(tile_digest(tile)), dtype=np.uint8)
            # pixels whose top-2 logit margin is below 2e-3 (packed bits): lets a test name the
            # razor-edge pixels where an fp32 reordering may legitimately flip the argmax
            top2 = torch.topk(logits[0], 2, dim=0).values
            out["edge_%d" % seed] = np.packbits((top2[0] - top2[1]).numpy() < 2e-3)
        save("masks_fold%d.npz" % fold, **out)

    net = nets[1]
    mean, std = FOLD_MEAN_STD[1]

    # (ii) logits on small tiles (fully convolutional: any size divisible by 8)
    small = {}
    for tag, (h, w, seed) in {"a": (64, 128, 100), "b": (128, 256, 101), "c": (72, 200, 102)}.items():
        tile = synth_tile(seed, h, w, blobs=6)
        small["tile_" + tag] = tile
        small["logits_" + tag] = net(preprocess(tile, mean, std))[0].numpy()
    save("small_fold1.npz", **small)

    # (iii) per-stage activations on the 64x128 tile
    stages = {}
    hooks = []

    def grab(name):
        def fn(_m, _i, o):
            stages[name] = o[0].numpy().copy()
        return fn

    enc = net.encoder
    named = {
        "level1": enc.level1, "sample1": enc.sample1, "sample2": enc.sample2, "b1": enc.b1,
        "level2_0": enc.level2_0, "level2.0": enc.level2[0], "level2.1": enc.level2[1], "b2": enc.b2,
        "level3_0": enc.level3_0, "b3": enc.b3, "enc_classifier": enc.classifier,
        "level3_0.c1": enc.level3_0.c1, "level2_0.c1": enc.level2_0.c1, "level3.0.c1": enc.level3[0].c1,
        "br": net.br, "up_l3": net.up_l3, "level3_C": net.level3_C,
        "combine_l2_l3": net.combine_l2_l3, "up_l2": net.up_l2, "conv": net.conv, "classifier": net.classifier,
    }
    for i in range(8):
        named["level3.%d" % i] = enc.level3[i]
    for k, m in named.items():
        hooks.append(m.register_forward_hook(grab(k)))
    tile = small["tile_a"]
    x = preprocess(tile, mean, std)
    logits = net(x)
    for hnd in hooks:
        hnd.remove()
    stages["input"] = x[0].numpy()
    stages["tile"] = tile
    stages["logits"] = logits[0].numpy()
    save("stages_fold1.npz", **stages)

    # (v) single-block known-answer tests with the real fold-1 weights; ragged sizes on purpose
    rng = np.random.default_rng(7)
    blocks = {}

    def kat(tag, module, shape):
        xin = rng.standard_normal(shape).astype(np.float32)
        blocks[tag + "_in"] = xin
        blocks[tag + "_out"] = module(torch.from_numpy(xin).unsqueeze(0))[0].numpy()

    kat("esp3", enc.level3[3], (128, 24, 40))          # d=16 zero padding fully exercised
    kat("esp2", enc.level2[1], (64, 20, 72))
    kat("down3", enc.level3_0, (131, 48, 80))
    kat("down2", enc.level2_0, (19, 40, 144))
    save("blocks_fold1.npz", **blocks)

    # ESPNet-C (modelType 2) path: encoder logits at 1/8 scale, same weights (encoder.* keys)
    encnet = RefModel.ESPNet_Encoder(5, 2, 8)
    sd = torch.load(os.path.join(REF, "models", "espnet_fold1.pth"), map_location="cpu")
    encnet.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    encnet.eval()
    save("encoder_fold1.npz", tile=small["tile_b"],
         out=encnet(preprocess(small["tile_b"], mean, std))[0].numpy())

    # cfg 5: 5-fold ensemble.  The reference has no ensemble code; the build
    # defines it as softmax-mean over folds, each fold with its own mean/std.  The same formula
    # run through the reference models pins the arithmetic.
    ens = {}
    for seed in (0, 1):
        tile = synth_tile(seed, 256, 512, blobs=8)
        prob = 0
        for fold in range(1, 6):
            m, s = FOLD_MEAN_STD[fold]
            prob = prob + torch.softmax(nets[fold](preprocess(tile, m, s))[0], dim=0)
        prob = prob / 5.0
        ens["tile_%d" % seed] = tile
        ens["mask_%d" % seed] = prob.max(0)[1].byte().numpy()
        top2 = torch.topk(prob, 2, dim=0).values
        ens["edge_%d" % seed] = np.packbits((top2[0] - top2[1]).numpy() < 1e-3)
    save("ensemble.npz", **ens)


if __name__ == "__main__":
    main()
