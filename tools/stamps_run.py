#!/usr/bin/env python3
"""Runs four 32-tile forwards on the library under GLOMSEG_LIB (a -DGS_DIAG build) with GS_VARIANT set by the caller, so that the
stamped launch of that variant (160..165, csrc/espnet_diag.inc) writes its gpurun_out/stamps_*.txt on the third call.
    GS_VARIANT=160 GLOMSEG_EXPERIMENT=1 GLOMSEG_ALLOW_DIAG=1 GLOMSEG_LIB=variants_so/libglomseg_diag.so python tools/stamps_run.py"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    big = torch.from_numpy(np.stack([synth_tile(s) for s in range(n)])).cuda()
    for _ in range(4):
        mask, hist, _ = eng.segment(big, mean, std)
        torch.cuda.synchronize()
    print("variant", os.environ.get("GS_VARIANT"), "counts", hist.sum(0).tolist())


if __name__ == "__main__":
    main()
