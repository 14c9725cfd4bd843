import os, sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD
mean, std = FOLD_MEAN_STD[1]
torch.cuda.set_device(0)
sd = bench.load_weights()
tiles_np = bench.make_batches(0)
dev = torch.device("cuda", 0)
tiles = torch.from_numpy(tiles_np).to(dev)
for NS in (1, 2, 3):
    engs = [EspnetEngine(sd) for _ in range(NS)]
    for e in engs: e.reserve(32, 512, 1024)
    streams = [torch.cuda.Stream() for _ in range(NS)]
    masks = [torch.zeros((4, 32, 512, 1024), dtype=torch.uint8, device=dev) for _ in range(NS)]
    hists = [torch.empty((32, 5), dtype=torch.int64, device=dev) for _ in range(NS)]
    totals = [torch.zeros(5, dtype=torch.int64, device=dev) for _ in range(NS)]
    cnt = [0]
    def step():
        i = cnt[0]; cnt[0] += 1
        k = i % NS
        with torch.cuda.stream(streams[k]):
            engs[k].segment(tiles[i % 4], mean, std, out_mask=masks[k][i % 4], out_hist=hists[k])
            totals[k].add_(hists[k].sum(0))
    for _ in range(8): step()
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(24): step()
        torch.cuda.synchronize(); res.append(time.perf_counter() - t0)
    el = float(np.median(res))
    print("streams %d: %.1f patches/s  %.3f ms/step" % (NS, 24 * 32 / el, el / 24 * 1e3), flush=True)
    for e in engs: e.close()
