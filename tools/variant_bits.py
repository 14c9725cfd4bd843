#!/usr/bin/env python3
"""SHA-256 of what the library under GLOMSEG_LIB computes on fixed inputs -- logits of the four golden tiles (batch 4 and batch 1),
masks and counts of a 32-tile batch at 1024x512 -- so that an experiment variant can be compared with the shipped build BIT FOR BIT:
    python tools/variant_bits.py                                   (the shipped library)
    GLOMSEG_EXPERIMENT=1 GLOMSEG_LIB=variants_so/x.so python tools/variant_bits.py
Equal digests = the same bits."""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from glomeruli_segmentation_amd.engine import EspnetEngine
    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile
    z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
    eng = EspnetEngine({k: z[k] for k in z.files})
    mean, std = FOLD_MEAN_STD[1]
    h = hashlib.sha256()
    small = torch.from_numpy(np.stack([synth_tile(s, 64, 128, blobs=4) for s in range(4)])).cuda()
    for t in (small, small[:1]):
        mask, hist, logits = eng.segment(t, mean, std, want_logits=True)
        for x in (mask, hist, logits):
            h.update(x.cpu().numpy().tobytes())
    big = torch.from_numpy(np.stack([synth_tile(s) for s in range(32)])).cuda()
    mask, hist, _ = eng.segment(big, mean, std)
    h.update(mask.cpu().numpy().tobytes())
    h.update(hist.cpu().numpy().tobytes())
    mask1, hist1, lg1 = eng.segment(big[:2], mean, std, want_logits=True)
    h.update(lg1.cpu().numpy().tobytes())
    print(os.environ.get("GLOMSEG_LIB", "shipped"), h.hexdigest())


if __name__ == "__main__":
    main()
