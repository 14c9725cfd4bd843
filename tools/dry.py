"""--dry-run stand-ins shared by tools/bench_slide.py and tools/bench_ensemble.py: objects with the interfaces of EspnetEngine /
FrcnnDetector / SlideCompositor / engine.segment_crops_host that compute a deterministic function of their inputs on the CPU, so
that the N-rank control flow of those tools (spawn, rank ranges, empty ranges, reductions, the JSON line) can be rehearsed with
gloo on a box without a GPU.  Never a measurement; nothing here is imported by the product package."""
import numpy as np
import torch


def fake_class_map(crop_bgr, classes=5):
    """a class map that depends on every pixel of the crop (so that a rank that got the wrong crop shows in the totals)"""
    c = np.asarray(crop_bgr)
    return ((c[:, :, 0].astype(np.int32) + 2 * c[:, :, 1] + 3 * c[:, :, 2]) // 37 % classes).astype(np.uint8)


class DryCompositor:
    """composite.SlideCompositor's surface on a CPU tensor: 1/8 map, max-composite (eval_wsi_segmentation.py:311-312)"""

    def __init__(self, width, height, device=None, ds=8):
        self.ds = ds
        self.map = torch.zeros(((height + ds - 1) // ds, (width + ds - 1) // ds), dtype=torch.uint8)

    def paste_target(self):
        return self

    def paste(self, mask, x1, y1):
        ds = self.ds
        m = self.map.numpy()
        h, w = mask.shape
        X0, Y0 = -(-x1 // ds), -(-y1 // ds)
        xs = np.arange(X0, min((x1 + w - 1) // ds + 1, m.shape[1]))
        ys = np.arange(Y0, min((y1 + h - 1) // ds + 1, m.shape[0]))
        if len(xs) and len(ys):
            sub = mask[np.ix_(ys * ds - y1, xs * ds - x1)]
            view = m[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1]
            np.maximum(view, sub, out=view)


def dry_segment_crops_host(engines, mean_stds, crops, net_h=512, net_w=1024, batch=32, want_masks=True, want_net_maps=False,
                           want_hist=True, paste=None, origins=None, overlay=None):
    classes = engines[0].classes
    masks = [fake_class_map(c, classes) for c in crops]
    if len(engines) > 1:      # an "ensemble": every member votes with a shifted map, the maximum wins (any deterministic rule will do)
        masks = [np.maximum(m, (m + len(engines)) % classes) for m in masks]
    if paste is not None:
        for m, (x1, y1) in zip(masks, origins):
            paste.paste(m, int(x1), int(y1))
    counts = np.array([np.bincount(m.ravel(), minlength=classes)[:classes] for m in masks], dtype=np.int64).reshape(len(masks), classes)
    return {"masks": masks if want_masks else None, "net_maps": None, "counts": counts, "overlays": None}


class DryEngine:
    def __init__(self, classes=5):
        self.classes = classes
        self.device = torch.device("cpu")
        self.encoder_only = False

    def segment_crops(self, crops, mean, std, net_h=512, net_w=1024, batch=32, **kw):
        return dry_segment_crops_host([self], [(mean, std)], crops, net_h, net_w, batch, **kw)

    def close(self):
        pass


class DryDetector:
    """the detect_box contract (detect_glomus_test.py:349-352): one box per window, placed by the window's mean colour"""

    def __call__(self, ims):
        ims = np.asarray(ims)
        n = len(ims)
        b = np.zeros((n, 1, 4), np.float32)
        for i in range(n):
            t = float(ims[i, ::64, ::64].mean()) / 255.0
            b[i, 0] = [0.2 * t, 0.3 * t, 0.2 * t + 0.3, 0.3 * t + 0.3]
        return b, np.full((n, 1), 0.9, np.float32), np.ones((n, 1), np.float32), np.ones((n,), np.float32)

    def close(self):
        pass
