"""Small-batch latency / throughput of the resident forward (the drop-in boundary's batch-1 call,
module/espnet/test/VisualizeResults_iou.py:119-123, and the 7-crops-per-rank shapes of BASELINE cfg 4 / 5 on 8 GPUs).

    python tools/latency.py [--out profiles/r04_latency.json] [--sizes 1,2,4,8,16,32]

Per batch size: ms per call with calls enqueued back to back (throughput), ms per call with a synchronise after every call
(latency), host time to enqueue one call, and the per-kernel average durations (HIP events around every launch)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from glomeruli_segmentation_amd.engine import EspnetEngine  # noqa: E402
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--sizes", default="1,2,4,8,16,32")
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
    eng = EspnetEngine({k: z[k] for k in z.files})
    mean, std = FOLD_MEAN_STD[1]
    rows = []
    for n in [int(v) for v in args.sizes.split(",")]:
        tiles = torch.from_numpy(np.stack([synth_tile(k) for k in range(n)])).cuda()
        mask = torch.empty((n, 512, 1024), dtype=torch.uint8, device="cuda")
        hist = torch.empty((n, 5), dtype=torch.int64, device="cuda")
        for _ in range(5):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        torch.cuda.synchronize()
        reps = args.reps
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        t_enq = (time.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(30):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
            torch.cuda.synchronize()
        t_lat = (time.perf_counter() - t0) / 30
        eng.profile(True)
        for _ in range(20):
            eng.segment(tiles, mean, std, out_mask=mask, out_hist=hist)
        kern = {k["name"]: {"avg_ms": round(k["total_ms"] / max(k["launches"], 1), 5), "launches_per_call": k["launches"] // 20}
                for k in eng.profile_read() if k["launches"]}
        eng.profile(False)
        row = {"tiles": n, "ms_per_call": round(t_all * 1e3, 4), "tiles_per_s": round(n / t_all, 1),
               "latency_ms": round(t_lat * 1e3, 4), "enqueue_ms": round(t_enq * 1e3, 4),
               "kernel_sum_ms": round(sum(v["avg_ms"] * v["launches_per_call"] for v in kern.values()), 4), "kernels": kern}
        rows.append(row)
        print("n=%2d  %.3f ms/call (%.0f tiles/s)  latency %.3f ms  enqueue %.3f ms  kernels %.3f ms" %
              (n, row["ms_per_call"], row["tiles_per_s"], row["latency_ms"], row["enqueue_ms"], row["kernel_sum_ms"]), flush=True)
        print("      " + "  ".join("%s=%.4f" % (k.replace("conv_", "").replace("_kernel", ""), v["avg_ms"]) for k, v in kern.items()),
              flush=True)
    eng.close()
    # the LITERAL drop-in: the reference's loop body over the import-swapped module -- float32 [1,3,H,W] on
    # the GPU in, `model(x)`, then VisualizeResults_iou.py:128 `img_out[0].max(0)[1].byte().cpu().data.numpy()`
    import glomeruli_segmentation_amd.Model as Net
    model = Net.ESPNet(5, 2, 8)
    model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
    model = model.to("cuda:0").eval()
    x = torch.rand((1, 3, 512, 1024), dtype=torch.float32, device="cuda:0") * 0.02 - 0.01
    for _ in range(5):
        model(x)[0].max(0)[1].byte().cpu().data.numpy()
    t0 = time.perf_counter()
    for _ in range(50):
        cm = model(x)[0].max(0)[1].byte().cpu().data.numpy()
    t_shim = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        lg = model(x)
    torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / 50
    shim = {"ms_per_image_with_argmax_and_copy_out": round(t_shim * 1e3, 4), "images_per_s": round(1.0 / t_shim, 1),
            "ms_per_forward_logits_only": round(t_fwd * 1e3, 4),
            "what": "glomeruli_segmentation_amd.Model.ESPNet called as the reference's loop calls its model: batch 1, float32 NCHW in, "
                    "fp32 logits out (10.5 MB written per tile), torch max + byte + copy to the host per image"}
    print("shim batch 1: %.3f ms per image incl. argmax + .cpu() (%.0f images/s); forward alone %.3f ms" %
          (shim["ms_per_image_with_argmax_and_copy_out"], shim["images_per_s"], shim["ms_per_forward_logits_only"]), flush=True)
    out = {"what": "resident forward, uint8 1024x512 tiles -> masks + counts, one lane, fold-1 weights", "rows": rows,
           "import_swap_batch1": shim}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
