"""Where the host pipeline (pinned tiles in -> pinned masks out) loses against the resident rate: raw copy rates, the
pipeline at several batch sizes / lane counts, and compute with copies running beside it."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD
mean, std = FOLD_MEAN_STD[1]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
sd = bench.load_weights()
t32 = bench.make_batches(0).reshape(-1, 512, 1024, 3)[:64]
host = torch.from_numpy(np.concatenate([t32] * 6)).pin_memory()          # 384 tiles
om = torch.zeros(host.shape[:3], dtype=torch.uint8).pin_memory()
oh = torch.zeros((host.shape[0], 5), dtype=torch.int64).pin_memory()
d = torch.empty_like(host[:32], device=dev)
torch.cuda.synchronize()
for nb in (1, 4):
    t0 = time.perf_counter()
    for _ in range(nb * 5):
        d.copy_(host[:32], non_blocking=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("H2D 50 MB x%d: %.1f GB/s" % (nb * 5, nb * 5 * d.numel() / el / 1e9), flush=True)
dm = torch.zeros((32, 512, 1024), dtype=torch.uint8, device=dev)
t0 = time.perf_counter()
for _ in range(10):
    om[:32].copy_(dm, non_blocking=True)
torch.cuda.synchronize()
print("D2H 16.8 MB x10: %.1f GB/s" % (10 * dm.numel() / (time.perf_counter() - t0) / 1e9), flush=True)
for lanes in (1, 2):
    eng = EspnetEngine(sd, lanes=lanes)
    for batch in (16, 32, 64):
        eng.segment_host(host[:4 * batch], mean, std, batch=batch, out_masks=om[:4 * batch], out_hist=oh[:4 * batch])
        t0 = time.perf_counter()
        eng.segment_host(host, mean, std, batch=batch, out_masks=om, out_hist=oh)
        el = time.perf_counter() - t0
        print("lanes %d batch %d: %.1f patches/s" % (lanes, batch, host.shape[0] / el), flush=True)
    eng.close()
