"""Pinned-host pipeline rate over 24 batches (argv[1] = lanes), unprofiled; with a GS_DIAG library GS_PIPE_SKIP=1/2/3 drops the
uploads / downloads / both (timing only)."""
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
host = torch.from_numpy(np.concatenate([t] * 6)).pin_memory()      # 768 tiles
om = torch.zeros(host.shape[:3], dtype=torch.uint8).pin_memory()
oh = torch.zeros((host.shape[0], 5), dtype=torch.int64).pin_memory()
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
eng = EspnetEngine(sd, lanes=lanes)
eng.segment_host(host[:128], mean, std, batch=32, out_masks=om[:128], out_hist=oh[:128])
for rep in range(3):
    t0 = time.perf_counter()
    eng.segment_host(host, mean, std, batch=32, out_masks=om, out_hist=oh)
    el = time.perf_counter() - t0
    print("lanes %d: %.1f patches/s  (%.3f ms per batch)" % (lanes, host.shape[0] / el, el / 24 * 1e3), flush=True)
