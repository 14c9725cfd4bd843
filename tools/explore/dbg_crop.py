import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from glomeruli_segmentation_amd.detector import FrcnnDetector, synthetic_weights
from oracle import detector_oracle as do
torch.cuda.set_device(0)
sd = synthetic_weights(0)
rng = np.random.default_rng(11)
H, W = 160, 192
imgs = rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8)
imgs[1, 40:120, 50:150] = (imgs[1, 40:120, 50:150] // 4 + 180).astype(np.uint8)
det = FrcnnDetector(sd)
out = {k: v.cpu().numpy() for k, v in det.forward_device(torch.from_numpy(imgs).cuda(), taps=True).items()}
for i in range(2):
    prop = out["proposals"][i]
    nv = int((np.abs(prop).sum(1) > 0).sum())
    head = do.box_head(out["features"][i], prop, H, W, sd)
    got = out["head"][i * 300:(i + 1) * 300]
    d = np.abs(got[:nv] - head[:nv]).max(1)
    bad = np.nonzero(d > 1e-4)[0]
    print(i, nv, bad[:10], d[bad][:10])
    for b in bad[:5]:
        print("   box", prop[b], "norm", prop[b] / np.array([H, W, H, W]))
