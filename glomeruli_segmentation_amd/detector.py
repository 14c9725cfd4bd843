"""The assembled detector behind the reference's ``detect_box`` tensor contract
(module/faster-rcnn/detect_glomus_test.py:349-352, tensors :443-450), running on the GPU through
``gs_detector_*`` (csrc/detector.hip).

The reference's network is an external TensorFlow-1.12 frozen graph (:419-427) whose architecture and weights are
not in the reference, so this is a detector of the same SHAPE with caller-supplied weights; ``synthetic_weights``
makes seeded ones (there is no network to download trained ones from).  Parity with the reference's graph is
unpinned; the graph itself is checked against oracle/detector_oracle.py.

    det = FrcnnDetector(synthetic_weights(seed=0))
    boxes, scores, classes, num = det(window_rgb_u8[None])        # the sess.run of :350-352
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .engine import pack_state_dict

# name -> (kernel, cin, cout); weights are [k,k,cin,cout] (TF layout), biases [cout]
LAYERS = {
    "backbone.c1": (3, 16, 64), "backbone.c2": (3, 64, 64), "backbone.c3": (3, 64, 128), "backbone.c4": (3, 128, 128),
    "backbone.c5": (3, 128, 256), "backbone.c6": (3, 256, 256), "rpn.conv": (3, 256, 256), "rpn.head": (1, 256, 72),
    "head.h1": (1, 256, 128), "head.h2": (3, 128, 128), "head.fc": (1, 128, 6),
}
ANCHORS_PER_CELL = 12
PROPOSALS = 300
MAX_DETECTIONS = 100


def synthetic_weights(seed=0):
    """Seeded random weights of the right shapes: He-normal for the ReLU layers, small heads (as detection heads are
    initialised), so activations stay O(1) through the stack and box deltas stay moderate."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, (k, cin, cout) in LAYERS.items():
        head = name in ("rpn.head", "head.fc")
        std = 0.05 if head else float(np.sqrt(2.0 / (k * k * cin)))
        w = rng.standard_normal((k, k, cin, cout)).astype(np.float32) * np.float32(std)
        if name == "backbone.c1":
            w[:, :, 12:, :] = 0.0          # the four padding channels of the space-to-depth input carry no weight
        sd[name + ".weight"] = w
        sd[name + ".bias"] = (rng.standard_normal(cout) * (0.5 if head else 0.05)).astype(np.float32)
    return sd


class FrcnnDetector:
    """One detector on one GPU.  Callable with the ``detect_box`` contract: uint8 RGB [N,H,W,3] (numpy or a GPU tensor)
    -> (boxes [N,D,4] normalised [ymin,xmin,ymax,xmax], scores [N,D] descending, classes [N,D], num [N]) as numpy."""

    def __init__(self, weights, device=None, rpn_nms_iou=0.7, det_nms_iou=0.6, score_threshold=0.0):
        if not torch.cuda.is_available():
            raise RuntimeError("FrcnnDetector needs a HIP device; there is no CPU path in this build")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        blob, table = pack_state_dict(weights)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_detector_create(blob.ctypes.data_as(ctypes.c_void_p), table, len(table), ctypes.byref(h)))
            self.handle = h
            _lib.check(self.lib.gs_detector_set_thresholds(h, rpn_nms_iou, det_nms_iou, score_threshold))
        self.max_det = self.lib.gs_detector_max_detections()

    def close(self):
        if getattr(self, "handle", None):
            with torch.cuda.device(self.device):
                self.lib.gs_detector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward_device(self, images, taps=False):
        """images: uint8 [N,H,W,3] GPU tensor -> dict of GPU tensors (boxes, scores, classes, num [+ debug taps])."""
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3 or not images.is_cuda:
            raise ValueError("expected a uint8 [N,H,W,3] tensor on the GPU")
        images = images.contiguous()
        n, h, w, _ = images.shape
        dev = images.device
        d = self.max_det
        out = {"boxes": torch.empty((n, d, 4), dtype=torch.float32, device=dev), "scores": torch.empty((n, d), dtype=torch.float32, device=dev),
               "classes": torch.empty((n, d), dtype=torch.float32, device=dev), "num": torch.empty((n,), dtype=torch.float32, device=dev)}
        dbg = [None, None, None, None]
        if taps:
            def down(x, k, s, p):
                return (x + 2 * p - k) // s + 1
            h2, w2 = (h + 1) // 2, (w + 1) // 2
            hf, wf = down(down(down(h2, 3, 2, 1), 3, 2, 1), 3, 2, 1), down(down(down(w2, 3, 2, 1), 3, 2, 1), 3, 2, 1)
            out["features"] = torch.empty((n, hf, wf, 256), dtype=torch.float32, device=dev)
            out["rpn"] = torch.empty((n, hf, wf, 72), dtype=torch.float32, device=dev)
            out["proposals"] = torch.empty((n, PROPOSALS, 4), dtype=torch.float32, device=dev)
            out["head"] = torch.empty((n * PROPOSALS, 6), dtype=torch.float32, device=dev)
            dbg = [out[k].data_ptr() for k in ("features", "rpn", "proposals", "head")]
        with torch.cuda.device(dev):
            _lib.check(self.lib.gs_detector_forward(
                self.handle, images.data_ptr(), n, h, w, out["boxes"].data_ptr(), out["scores"].data_ptr(), out["classes"].data_ptr(),
                out["num"].data_ptr(), dbg[0], dbg[1], dbg[2], dbg[3], ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out

    def detect_host(self, windows, batch=16):
        """gs_detector_detect_host: a list of equal-size uint8 RGB [H,W,3] windows in host memory (numpy; pinned CPU tensors are
        DMA'd in place) -> (boxes [n,D,4], scores [n,D], classes [n,D], num [n]) numpy.  Uploads run one batch ahead of the
        forward on a stream of their own; pageable windows are staged into pinned memory by a few threads."""
        n = len(windows)
        keep, ptrs = [], (ctypes.c_void_p * n)()
        h = w = None
        for i, im in enumerate(windows):
            if isinstance(im, torch.Tensor):
                if im.is_cuda or im.dtype != torch.uint8 or not im.is_contiguous():
                    raise ValueError("window %d: expected a contiguous uint8 CPU tensor" % i)
                ptrs[i] = im.data_ptr()
            else:
                im = np.ascontiguousarray(im, dtype=np.uint8)
                ptrs[i] = im.ctypes.data
            if im.ndim != 3 or im.shape[2] != 3 or (h is not None and tuple(im.shape[:2]) != (h, w)):
                raise ValueError("window %d: expected uint8 [H,W,3], every window of one size" % i)
            h, w = int(im.shape[0]), int(im.shape[1])
            keep.append(im)
        d = self.max_det
        boxes = np.empty((n, d, 4), dtype=np.float32)
        scores = np.empty((n, d), dtype=np.float32)
        classes = np.empty((n, d), dtype=np.float32)
        num = np.empty((n,), dtype=np.float32)
        torch.cuda.current_stream(self.device).synchronize()     # the pipeline runs on streams of its own
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_detector_detect_host(self.handle, ptrs, n, h, w, int(batch), boxes.ctypes.data_as(ctypes.c_void_p),
                                                        scores.ctypes.data_as(ctypes.c_void_p), classes.ctypes.data_as(ctypes.c_void_p),
                                                        num.ctypes.data_as(ctypes.c_void_p)))
        return boxes, scores, classes, num

    def __call__(self, images):
        if not isinstance(images, torch.Tensor):
            images = torch.from_numpy(np.ascontiguousarray(images, dtype=np.uint8))
        out = self.forward_device(images.to(self.device))
        return (out["boxes"].cpu().numpy(), out["scores"].cpu().numpy(), out["classes"].cpu().numpy(), out["num"].cpu().numpy())
