"""gs_espnet_segment_host from PAGEABLE caller buffers (numpy arrays, as the reference's loop has them) vs pinned ones."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD
mean, std = FOLD_MEAN_STD[1]
torch.cuda.set_device(0)
sd = bench.load_weights()
t = bench.make_batches(0).reshape(-1, 512, 1024, 3)
tiles = np.concatenate([t] * 4)        # 512 tiles, pageable
eng = EspnetEngine(sd, lanes=2)
eng.segment_host(tiles[:128], mean, std, batch=32)
for rep in range(3):
    t0 = time.perf_counter()
    m, h = eng.segment_host(tiles, mean, std, batch=32)
    el = time.perf_counter() - t0
    print("pageable: %.1f patches/s  (%.3f ms per batch)" % (tiles.shape[0] / el, el / 16 * 1e3), flush=True)
