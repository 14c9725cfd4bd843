#!/usr/bin/env python3
"""Golden vectors for two small pieces of the per-patch driver, produced by the REFERENCE's own code:
  * module/common/IOUEval.py: iouEval.addBatch / getMetricRight (imports torch and numpy only);
  * module/espnet/test/VisualizeResults_iou.py: relabel (the --cityFormat cascade, :54-81).  That script imports cv2 and labelme
    (not installed) at module level; relabel touches neither, so empty placeholder modules are registered under those names
    for the import only.
Only arrays are written.   python tests/golden/make_golden_misc.py  ->  tests/golden/misc.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GS_REFERENCE", "/root/reference")
for name in ("cv2", "labelme", "labelme.utils"):
    sys.modules.setdefault(name, types.ModuleType(name))                 # import-time placeholders only
sys.modules["labelme"].utils = sys.modules["labelme.utils"]
sys.path.insert(0, os.path.join(REF, "module", "common"))
sys.path.insert(0, os.path.join(REF, "module", "espnet", "test"))
from IOUEval import iouEval  # noqa: E402
import VisualizeResults_iou as ref  # noqa: E402

out = {}
rng = np.random.default_rng(5)
ev = iouEval(5)
for k in range(3):
    pred = rng.integers(0, 5, (40, 50)).astype(np.int64)
    gt = rng.integers(0, 5 if k < 2 else 4, (40, 50)).astype(np.uint8)      # the last pair has no class 4 in the ground truth
    if k == 1:
        pred[gt == 2] = 2                                                      # one class perfectly predicted
    hist = ev.addBatch(pred, gt)
    out["pred_%d" % k], out["gt_%d" % k], out["hist_%d" % k] = pred, gt, hist
o, pa, pi, m = ev.getMetricRight()
out["total_hist"], out["overall_acc"], out["per_class_acc"], out["per_class_iu"], out["miou"] = ev.hist, o, pa, pi, m
out["relabel_in"] = np.arange(256, dtype=np.uint8)
out["relabel_out"] = ref.relabel(np.arange(256, dtype=np.uint8))
out["palette"] = np.array(ref.pallete, dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "misc.npz"), **out)
print(o, m, out["relabel_out"][:21])
