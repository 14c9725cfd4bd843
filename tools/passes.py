"""diagnostic: alternate un-instrumented and instrumented timed passes of the bench step"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile
z = np.load("tests/golden/weights_fold1.npz"); sd = {k: z[k] for k in z.files}
mean, std = FOLD_MEAN_STD[1]
eng = EspnetEngine(sd); B = 32
tiles = torch.from_numpy(np.stack([synth_tile(i) for i in range(B)])).cuda()
mask = torch.empty((B, 512, 1024), dtype=torch.uint8, device="cuda"); hist = torch.empty((B, 5), dtype=torch.int64, device="cuda")
def run(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for _ in range(3): run(1)
for i in range(8):
    prof = i % 2 == 1
    eng.profile(prof)
    ms = run(20)
    if prof: eng.profile_read()
    print("pass %d profiled=%s ms/step %.3f" % (i, prof, ms), flush=True)
