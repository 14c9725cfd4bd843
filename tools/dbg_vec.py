import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile
z = np.load("tests/golden/weights_fold1.npz"); sd = {k: z[k] for k in z.files}
mean, std = FOLD_MEAN_STD[1]
eng = EspnetEngine(sd)
H, W = int(sys.argv[1]), int(sys.argv[2])
tiles = torch.from_numpy(np.stack([synth_tile(i, H, W) for i in range(2)])).cuda()
names = ["b1", "level2_0", "level2.0", "level2.1", "b2", "level3_0", "level3.0", "level3.7", "up_l3", "up_l2", "conv"]
res = {}
for mode in ("1", "0"):
    os.environ["GS_NO_VEC"] = mode
    eng.segment(tiles, mean, std)
    torch.cuda.synchronize()
    for nm in names:
        try:
            res[(mode, nm)] = eng.read_stage(nm, 1)
        except Exception as e:
            pass
for nm in names:
    if ("1", nm) in res and ("0", nm) in res:
        a, b = res[("1", nm)], res[("0", nm)]
        d = np.abs(a - b)
        idx = np.unravel_index(d.argmax(), d.shape)
        bad = np.argwhere(d > 1e-3)
        print(nm, a.shape, "max diff", d.max(), "at", idx, "nbad", len(bad), "first", bad[:3].tolist(), "xs", sorted(set(bad[:, 2].tolist()))[:20] if len(bad) else "")
a, b = res[("1", "level2_0")], res[("0", "level2_0")]
d = np.abs(a - b); bad = np.argwhere(d > 1e-3)
print("channels", sorted(set(bad[:, 0].tolist())))
print("rows", sorted(set(bad[:, 1].tolist())))
for c, y, x in bad[:12].tolist():
    print(c, y, x, "novec", a[c, y, x], "vec", b[c, y, x], "diff", b[c, y, x] - a[c, y, x])
