#!/usr/bin/env python3
"""Secondary measurements on one MI355X (not the headline; bench.py owns that): BASELINE configs 3-5.

  cfg 3  detector primitives on 16 synthetic 1000x1000x3 windows: gs_conv2d_nhwc (first backbone-style
         layers), gs_roialign (300 boxes -> 14x14 crops), gs_nms (300 boxes)  [SURVEY 8d: report GB/s],
         and the assembled detector forward (gs_detector_forward) on the same sixteen windows
  cfg 4  crop stage (bilinear resize + normalise of 1098^2 crops to 1024x512), nearest resize back,
         WSI max-compositor paste
  cfg 5  5-fold ensemble (softmax mean over the five shipped folds) on a batch of 32 tiles

Prints one JSON object; `python tools/bench_aux.py > profiles/rNN_aux_bench.json` on the GPU box.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from glomeruli_segmentation_amd import _lib  # noqa: E402
from glomeruli_segmentation_amd.engine import EspnetEngine, crop_preprocess, ensemble_segment, mask_resize_nearest  # noqa: E402
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {}
    g = torch.Generator().manual_seed(0)

    # ---- cfg 3: detector primitives -------------------------------------------------------------
    n, h, w = 16, 1000, 1000
    x = torch.rand(n, h, w, 3, generator=g).to(dev)
    convs = []
    cin = 3
    cur = x
    for cout, stride in ((64, 2), (64, 1), (128, 2)):
        wt = (torch.randn(3, 3, cin, cout, generator=g) * 0.1).to(dev)
        bs = torch.zeros(cout).to(dev)
        oh, ow = (cur.shape[1] + 2 - 3) // stride + 1, (cur.shape[2] + 2 - 3) // stride + 1
        y = torch.empty((n, oh, ow, cout), device=dev)
        inp = cur

        def run(inp=inp, wt=wt, bs=bs, y=y, cin=cin, cout=cout, stride=stride):
            _lib.check(lib.gs_conv2d_nhwc(inp.data_ptr(), n, inp.shape[1], inp.shape[2], cin, wt.data_ptr(), 3, 3, cout,
                                          bs.data_ptr(), stride, 1, 1, y.data_ptr(), None))
        t = timeit(run, reps=10)
        flops = 2.0 * n * oh * ow * cout * 9 * cin
        byts = 4.0 * (inp.numel() + y.numel())
        convs.append({"shape": "%dx%dx%d -> %d, 3x3 s%d" % (inp.shape[1], inp.shape[2], cin, cout, stride), "ms": round(t * 1e3, 3),
                      "TFLOP/s": round(flops / t / 1e12, 2), "GB/s": round(byts / t / 1e9, 1)})
        cur, cin = y, cout
    out["cfg3_conv2d_nhwc_batch16"] = convs

    feat = cur                                           # [16, 250, 250, 128]
    k = 300
    c = torch.rand(k, 2, generator=g) * 0.8
    s = torch.rand(k, 2, generator=g) * 0.2 + 0.02
    boxes = torch.cat([c, c + s], 1).to(dev)
    bimg = (torch.arange(k) % n).to(torch.int32).to(dev)
    crop = 14
    crops = torch.empty((k, crop, crop, feat.shape[3]), device=dev)
    t = timeit(lambda: _lib.check(lib.gs_roialign(feat.data_ptr(), n, feat.shape[1], feat.shape[2], feat.shape[3], boxes.data_ptr(),
                                                  bimg.data_ptr(), k, crop, crops.data_ptr(), None)))
    byts = crops.numel() * 4 * 5.0                       # 4 sampled reads + 1 write per output element
    out["cfg3_roialign_300x14x14x128"] = {"us": round(t * 1e6, 1), "GB/s": round(byts / t / 1e9, 1)}
    sc = torch.rand(k, generator=g).to(dev)
    keep = torch.full((k,), -1, dtype=torch.int32, device=dev)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    t = timeit(lambda: _lib.check(lib.gs_nms(boxes.data_ptr(), sc.data_ptr(), k, ctypes.c_float(0.6), ctypes.c_float(0.0), 100,
                                             keep.data_ptr(), nk.data_ptr(), None)))
    out["cfg3_nms_300"] = {"us": round(t * 1e6, 1), "pair_ious_per_s": round(k * k / t / 1e9, 3), "unit": "G pairs/s"}

    # ---- cfg 3: the assembled detector (gs_detector_forward), sixteen 1000x1000 windows per call ----------------
    from glomeruli_segmentation_amd.detector import LAYERS, FrcnnDetector, synthetic_weights
    det = FrcnnDetector(synthetic_weights(0))
    wins = torch.from_numpy(np.stack([synth_tile(200 + i, 1000, 1000, blobs=8)[:, :, ::-1].copy() for i in range(4)] * 4)).to(dev)
    t = timeit(lambda: det.forward_device(wins), reps=5, warm=2)
    sizes = {"backbone.c1": 500 * 500, "backbone.c2": 250 * 250, "backbone.c3": 125 * 125, "backbone.c4": 125 * 125, "backbone.c5": 63 * 63,
             "backbone.c6": 63 * 63, "rpn.conv": 63 * 63, "rpn.head": 63 * 63, "head.h1": 300 * 49, "head.h2": 300 * 16, "head.fc": 300}
    gflop = sum(2.0 * sizes[k] * kk * kk * ci * co for k, (kk, ci, co) in LAYERS.items()) / 1e9
    out["cfg3_detector_forward_batch16_1000x1000"] = {
        "ms_per_batch": round(t * 1e3, 2), "windows/s": round(16 / t, 1), "GFLOP_per_window": round(gflop, 2),
        "TFLOP/s": round(16 * gflop / t / 1e3, 1), "frac_of_fp32_mfma_peak": round(16 * gflop / t / 1e3 / 157.3, 3),
        "note": "synthetic weights (the reference's graph is external); uint8 windows resident in HBM -> detect_box tensors"}
    det.close()
    del wins

    # ---- cfg 4: crop stage + compositor ----------------------------------------------------------
    mean, std = FOLD_MEAN_STD[1]
    crop_u8 = torch.from_numpy(synth_tile(5, 1098, 1098)).to(dev)
    dst = torch.empty((3, 512, 1024), device=dev)
    t = timeit(lambda: crop_preprocess(crop_u8, mean, std, 512, 1024, out=dst))
    out["cfg4_crop_preprocess_1098sq_to_1024x512"] = {"us": round(t * 1e6, 1),
                                                      "GB/s": round((crop_u8.numel() + dst.numel() * 4) / t / 1e9, 1)}
    m = torch.randint(0, 5, (512, 1024), dtype=torch.uint8, device=dev)
    t = timeit(lambda: mask_resize_nearest(m, 1098, 1098))
    out["cfg4_mask_resize_nearest_to_1098sq"] = {"us": round(t * 1e6, 1)}
    from glomeruli_segmentation_amd.composite import SlideCompositor
    comp = SlideCompositor(40000, 40000, dev)
    big = mask_resize_nearest(m, 1098, 1098)
    t = timeit(lambda: comp.paste(big, 12345, 23456))
    out["cfg4_compositor_paste_1098sq"] = {"us": round(t * 1e6, 1)}

    # ---- cfg 4: the batched crop entry (gs_espnet_segment_crops, device-resident): 32 crops of the example slide's sizes
    z1 = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
    eng1 = EspnetEngine({kk: z1[kk] for kk in z1.files})
    ex = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
    sizes = [(int(b[3] - b[1]), int(b[2] - b[0])) for b in ex] + [(int(b[3] - b[1]), int(b[2] - b[0])) for b in ex[:4]]
    descs, ioff, ooff = [], 0, 0
    for (hh, ww) in sizes:
        d = _lib.CropDesc()
        d.in_off, d.out_off, d.h, d.w, d.x1, d.y1 = ioff, ooff, hh, ww, 8 * (ioff % 4000), 8 * (ooff % 3000)
        descs.append(d)
        ioff += hh * ww * 3
        ooff += (hh * ww + 255) // 256 * 256
    packed = torch.randint(0, 256, (ioff,), dtype=torch.uint8, device=dev)
    pout = torch.empty(ooff, dtype=torch.uint8, device=dev)
    from glomeruli_segmentation_amd.composite import SlideCompositor as SC
    comp2 = SC(40000, 40000, dev)
    pt = comp2.paste_target()
    t_all = timeit(lambda: eng1.segment_crops_resident(packed, descs, mean, std, 512, 1024, packed_out=pout, paste=pt), reps=10)
    tiles32 = torch.from_numpy(np.stack([synth_tile(i) for i in range(32)])).to(dev)
    t_u8 = timeit(lambda: eng1.segment(tiles32, mean, std), reps=10)
    out["cfg4_batched_crop_entry_32_crops"] = {
        "ms_per_batch": round(t_all * 1e3, 3), "crops/s": round(32 / t_all, 1),
        "same_engine_uint8_tiles_ms_per_batch": round(t_u8 * 1e3, 3),
        "overhead_vs_network_sized_uint8_tiles": round(t_all / t_u8, 3),
        "mean_crop_px": int(np.mean([a * b for a, b in sizes])),
        "note": "resample 32 crops (example-slide sizes) -> forward -> masks -> resize back + counts -> paste into a 5000 x 5000 map, "
                "everything resident in HBM, one lane; against the same engine on 32 network-sized uint8 tiles"}

    # ---- cfg 5: five-fold ensemble ---------------------------------------------------------------
    engines, mss = [], []
    for f in range(1, 6):
        z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold%d.npz" % f))
        engines.append(EspnetEngine({kk: z[kk] for kk in z.files}))
        mss.append(FOLD_MEAN_STD[f])
    tiles = torch.from_numpy(np.stack([synth_tile(i) for i in range(32)])).to(dev)
    t = timeit(lambda: ensemble_segment(engines, tiles, mss), reps=5, warm=2)
    out["cfg5_ensemble_5fold_batch32"] = {"ms_per_batch": round(t * 1e3, 2), "patches/s": round(32 / t, 1),
                                          "model_passes/s": round(5 * 32 / t, 1),
                                          "single_model_passes/s_same_run_one_lane": round(32 / t_u8, 1),
                                          "ratio_to_single_model_pass_rate": round(5 * t_u8 / t, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
