"""How much of a two-lane step is host-side bookkeeping around the forward call (stream wait, record_stream)?"""
import sys, time, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from glomeruli_segmentation_amd import _lib
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD
mean, std = FOLD_MEAN_STD[1]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
sd = bench.load_weights()
tiles = torch.from_numpy(bench.make_batches(0)).to(dev)
eng = EspnetEngine(sd, lanes=2)
eng.reserve(32, 512, 1024)
mask = torch.zeros((4, 32, 512, 1024), dtype=torch.uint8, device=dev)
hist = torch.zeros((24, 32, 5), dtype=torch.int64, device=dev)
lib = eng.lib
m3, s3 = _lib.fptr3(mean), _lib.fptr3(std)
streams = [eng.lane_stream(0), eng.lane_stream(1)]

def run(mode):
    def step(i):
        b, k = i % 4, i % 2
        if mode == "engine":
            eng.segment(tiles[b], mean, std, out_mask=mask[b], out_hist=hist[i % 24], lane=k)
        else:   # bare C call on the lane's stream: no wait_stream, no record_stream, no tensor checks
            _lib.check(lib.gs_espnet_forward_lane(eng.handle, k, tiles[b].data_ptr(), 0, 32, 512, 1024, m3, s3, None, mask[b].data_ptr(),
                                                  hist[i % 24].data_ptr(), ctypes.c_void_p(streams[k].cuda_stream)))
    for i in range(8): step(i)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(24): step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0, t1 - t0))
    el = sorted(res)[2]
    print("%-8s %.1f patches/s  %.3f ms/step   host enqueue %.3f ms/step" % (mode, 24 * 32 / el[0], el[0] / 24 * 1e3, el[1] / 24 * 1e3), flush=True)

for mode in ("engine", "bare", "engine", "bare"):
    run(mode)
