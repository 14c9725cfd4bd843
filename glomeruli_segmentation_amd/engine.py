"""Host-side owner of a ``gs_espnet`` handle: packs a reference state_dict into the weight blob
the C ABI takes and exposes the hot-path calls on torch HIP tensors.

torch is used for device memory and streams only; all arithmetic happens in libglomseg.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _to_numpy(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def pack_state_dict(state_dict):
    """state_dict (name -> tensor/array) -> (fp32 blob, LayerDesc array).  Integer buffers such as
    BatchNorm's num_batches_tracked are not arithmetic inputs and are skipped.
    Replaces torch.load + load_state_dict at VisualizeResults_iou.py:272,279."""
    items = []
    total = 0
    for name, v in state_dict.items():
        a = _to_numpy(v)
        if a.dtype.kind != "f":
            continue
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim > 4:
            raise ValueError("tensor %s has %d dims" % (name, a.ndim))
        items.append((name, a, total))
        total += a.size
    blob = np.empty(total, dtype=np.float32)
    table = (_lib.LayerDesc * len(items))()
    for i, (name, a, off) in enumerate(items):
        blob[off:off + a.size] = a.ravel()
        enc = name.encode()
        if len(enc) >= 96:
            raise ValueError("tensor name too long: " + name)
        table[i].name = enc
        table[i].offset = off
        table[i].ndim = a.ndim
        for d in range(a.ndim):
            table[i].shape[d] = a.shape[d]
    return blob, table


def _check_out(t, name, shape, dtype, device):
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() or \
            t.device.type != torch.device(device).type or (t.is_cuda and t.device != torch.device(device)):
        raise ValueError("%s must be a contiguous %s tensor of shape %s on %s" % (name, dtype, tuple(shape), device))


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class EspnetEngine:
    """One model on one GPU.  ``encoder_only`` builds ESPNet-C (keys without the 'encoder.' prefix)."""

    def __init__(self, state_dict, classes=5, p=2, q=8, encoder_only=False, device=None, lanes=1):
        if not torch.cuda.is_available():
            raise RuntimeError("EspnetEngine needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU path in this build")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   torch.device(device).index or 0)
        self.classes, self.p, self.q, self.encoder_only = classes, p, q, encoder_only
        blob, table = pack_state_dict(state_dict)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_create(blob.ctypes.data_as(ctypes.c_void_p), table, len(table), classes, p,
                                                 q, 1 if encoder_only else 0, ctypes.byref(h)))
        self.handle = h
        # lanes: extra activation workspaces (weights shared) so that several batches are in flight, each on a stream of
        # its own -- the tail of one batch's kernels overlaps the head of another's (include/glomseg.h, "LANES")
        self.lanes = 1
        self._lane_streams = {}
        if lanes != 1:
            self.set_lanes(lanes)

    def set_lanes(self, n):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_set_lanes(self.handle, int(n)))
        self.lanes = int(n)

    def lane_stream(self, lane):
        """the torch stream lane `lane` runs on (created on first use)"""
        if not 0 <= lane < self.lanes:
            raise ValueError("lane %d of %d" % (lane, self.lanes))
        if lane not in self._lane_streams:
            self._lane_streams[lane] = torch.cuda.Stream(device=self.device)
        return self._lane_streams[lane]

    def wait_lanes(self):
        """make the current stream wait for everything submitted to the lanes (their outputs are then safe to use on it)"""
        cur = torch.cuda.current_stream(self.device)
        for st in self._lane_streams.values():
            cur.wait_stream(st)

    def quiesce(self):
        """block until everything submitted for this handle -- on the current stream or on a lane's stream -- has finished.
        The host pipelines (segment_host, segment_crops) run on streams of the library's own in workspaces 0 and 1 and
        expect exactly that (include/glomseg.h)."""
        for st in self._lane_streams.values():
            st.synchronize()
        torch.cuda.current_stream(self.device).synchronize()

    def check_device_faults(self):
        """Synchronise the device and raise if a kernel of an earlier stream-ordered call (segment / forward_logits / lanes) reported
        through the device-side fault word that it went on without data it was waiting for (gs_device_fault_check; the host
        pipelines check by themselves).  Cheap: one device synchronise and a 4-byte read."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_device_fault_check())

    def close(self):
        if getattr(self, "handle", None):
            with torch.cuda.device(self.device):
                self.lib.gs_espnet_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ hot path
    def reserve(self, n, height, width):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_reserve(self.handle, n, height, width))

    def forward_logits(self, x):
        """fp32 [N,3,H,W] on the GPU -> logits [N,classes,H,W] ([N,classes,H/8,W/8] for ESPNet-C).
        The nn.Module.forward replacement (Model.py:341 / :273)."""
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda:
            raise ValueError("expected a float32 [N,3,H,W] tensor on the GPU")
        x = x.contiguous()
        n, _, h, w = x.shape
        oh, ow = (h // 8, w // 8) if self.encoder_only else (h, w)
        out = torch.empty((n, self.classes, oh, ow), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.gs_espnet_forward(self.handle, x.data_ptr(), _lib.GS_IN_F32_NCHW, n, h, w, None, None,
                                                  out.data_ptr(), None, None, _stream_ptr(x.device)))
        return out

    def segment(self, tiles_u8, mean, std, want_logits=False, want_hist=True, out_mask=None, out_hist=None, lane=None):
        """uint8 BGR [N,H,W,3] on the GPU -> (mask uint8 [N,H,W], counts int64 [N,classes], logits|None).
        One pass of VisualizeResults_iou.py:107-128,151-155 for a batch.
        lane=None: on the current stream, in workspace 0.  lane=k: in workspace k on that lane's own stream (which first
        waits for the current stream, so inputs produced there are ready); the outputs belong to the lane's stream until
        wait_lanes() -- submit the next batch to another lane meanwhile."""
        if tiles_u8.dtype != torch.uint8 or tiles_u8.dim() != 4 or tiles_u8.shape[3] != 3 or not tiles_u8.is_cuda:
            raise ValueError("expected a uint8 [N,H,W,3] tensor on the GPU")
        if self.encoder_only:
            raise ValueError("segment() needs the full ESPNet (decoder)")
        tiles_u8 = tiles_u8.contiguous()
        n, h, w, _ = tiles_u8.shape
        dev = tiles_u8.device
        # caller-supplied outputs are written by the kernels as plain pointers: refuse anything that is not exactly the
        # buffer the call would have allocated itself
        _check_out(out_mask, "out_mask", (n, h, w), torch.uint8, dev)
        _check_out(out_hist, "out_hist", (n, self.classes), torch.int64, dev)
        mask = out_mask if out_mask is not None else torch.empty((n, h, w), dtype=torch.uint8, device=dev)
        hist = None
        if want_hist:
            hist = out_hist if out_hist is not None else torch.empty((n, self.classes), dtype=torch.int64, device=dev)
        logits = torch.empty((n, self.classes, h, w), dtype=torch.float32, device=dev) if want_logits else None
        with torch.cuda.device(dev):
            if lane is None:
                if 0 in self._lane_streams:      # workspace 0 may still be in use on lane 0's own stream
                    torch.cuda.current_stream(dev).wait_stream(self._lane_streams[0])
                k, sp = 0, _stream_ptr(dev)
            else:
                st = self.lane_stream(lane)
                st.wait_stream(torch.cuda.current_stream(dev))
                for t in (tiles_u8, mask, hist, logits):      # (allocator: these tensors are in use on the lane's stream)
                    if t is not None:
                        t.record_stream(st)
                k, sp = lane, ctypes.c_void_p(st.cuda_stream)
            _lib.check(self.lib.gs_espnet_forward_lane(
                self.handle, k, tiles_u8.data_ptr(), _lib.GS_IN_U8_BGR_NHWC, n, h, w, _lib.fptr3(mean), _lib.fptr3(std),
                logits.data_ptr() if want_logits else None, mask.data_ptr(),
                hist.data_ptr() if want_hist else None, sp))
        return mask, hist, logits

    def segment_host(self, tiles, mean, std, batch=32, want_hist=True, out_masks=None, out_hist=None):
        """numpy uint8 [T,H,W,3] in host memory -> (masks [T,H,W], counts [T,classes]) through the
        pinned double-buffered H2D / compute / D2H pipeline of the library."""
        self.quiesce()
        if isinstance(tiles, torch.Tensor):      # e.g. a pinned CPU tensor: DMA'd in place, no staging copy
            if tiles.is_cuda or tiles.dtype != torch.uint8 or not tiles.is_contiguous():
                raise ValueError("expected a contiguous uint8 CPU tensor")
            t, h, w, _ = tiles.shape
            in_ptr = ctypes.c_void_p(tiles.data_ptr())
            pin = tiles.is_pinned()
            _check_out(out_masks, "out_masks", (t, h, w), torch.uint8, torch.device("cpu"))
            _check_out(out_hist, "out_hist", (t, self.classes), torch.int64, torch.device("cpu"))
            masks_t = out_masks if out_masks is not None else torch.empty((t, h, w), dtype=torch.uint8, pin_memory=pin)
            hist_t = None
            if want_hist:
                hist_t = out_hist if out_hist is not None else torch.zeros((t, self.classes), dtype=torch.int64, pin_memory=pin)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.gs_espnet_segment_host(
                    self.handle, in_ptr, t, h, w, _lib.fptr3(mean), _lib.fptr3(std), batch, ctypes.c_void_p(masks_t.data_ptr()),
                    ctypes.c_void_p(hist_t.data_ptr()) if want_hist else None))
            return masks_t.numpy(), (hist_t.numpy() if want_hist else None)
        tiles = np.ascontiguousarray(tiles, dtype=np.uint8)
        t, h, w, _ = tiles.shape
        masks = np.empty((t, h, w), dtype=np.uint8)
        hist = np.zeros((t, self.classes), dtype=np.uint64) if want_hist else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_segment_host(
                self.handle, tiles.ctypes.data_as(ctypes.c_void_p), t, h, w, _lib.fptr3(mean), _lib.fptr3(std), batch,
                masks.ctypes.data_as(ctypes.c_void_p), hist.ctypes.data_as(ctypes.c_void_p) if want_hist else None))
        return masks, (hist.astype(np.int64) if want_hist else None)

    def segment_crops(self, crops, mean, std, net_h=512, net_w=1024, batch=32, **kw):
        """The loop of VisualizeResults_iou.py:100-156 over crops of ANY sizes (numpy uint8 BGR [h,w,3] each, or pinned CPU
        tensors) through the library's batched pipeline; see segment_crops_host for the keywords and the result."""
        return segment_crops_host([self], [(mean, std)], crops, net_h, net_w, batch, **kw)

    def segment_crops_resident(self, packed_in, descs, mean, std, net_h=512, net_w=1024, want_net_maps=True, want_hist=True,
                               packed_out=None, paste=None, lane=None):
        """Device-resident form (gs_espnet_segment_crops): packed_in is a uint8 GPU tensor holding the crops at
        descs[i].in_off, descs a list of _lib.CropDesc.  Returns (net_maps [n,net_h,net_w] | None, counts [n,5] | None);
        crop-size maps are written into packed_out (uint8 GPU tensor) at descs[i].out_off when it is given."""
        n = len(descs)
        dev = packed_in.device
        tab = (_lib.CropDesc * n)(*descs)
        net = torch.empty((n, net_h, net_w), dtype=torch.uint8, device=dev) if want_net_maps else None
        hist = torch.empty((n, self.classes), dtype=torch.int64, device=dev) if want_hist else None
        with torch.cuda.device(dev):
            if lane is None:
                if 0 in self._lane_streams:
                    torch.cuda.current_stream(dev).wait_stream(self._lane_streams[0])
                k, sp = 0, _stream_ptr(dev)
            else:
                st = self.lane_stream(lane)
                st.wait_stream(torch.cuda.current_stream(dev))
                for t in (packed_in, packed_out, net, hist):
                    if t is not None:
                        t.record_stream(st)
                k, sp = lane, ctypes.c_void_p(st.cuda_stream)
            _lib.check(self.lib.gs_espnet_segment_crops(
                self.handle, k, packed_in.data_ptr(), tab, n, _lib.fptr3(mean), _lib.fptr3(std), net_h, net_w,
                net.data_ptr() if net is not None else None, packed_out.data_ptr() if packed_out is not None else None,
                hist.data_ptr() if hist is not None else None, ctypes.byref(paste) if paste is not None else None, sp))
        return net, hist

    # ------------------------------------------------------------------ test / bench hooks
    def read_stage(self, name, image=0):
        dims = (ctypes.c_int * 3)()
        _lib.check(self.lib.gs_espnet_read_stage(self.handle, name.encode(), image, None, 0, ctypes.byref(dims)))
        out = np.empty(tuple(dims), dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_read_stage(self.handle, name.encode(), image,
                                                     out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(dims)))
        return out

    def block_forward(self, kind, level, index, x):
        """One block of the trunk on a host CHW fp32 array (test hook; kind 0 = ESP block, 1 = DownSamplerB)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        c, h, w = x.shape
        cout = 64 if level == 2 else 128
        out = np.empty((cout, h, w) if kind == 0 else (cout, h // 2, w // 2), dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_block_forward(self.handle, kind, level, index, x.ctypes.data_as(ctypes.c_void_p),
                                                        h, w, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def profile(self, on):
        _lib.check(self.lib.gs_espnet_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        arr = (_lib.KernelTime * 32)()
        n = ctypes.c_int()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.gs_espnet_profile_read(self.handle, arr, 32, ctypes.byref(n)))
        return [{"name": arr[i].name.decode(), "total_ms": arr[i].total_ms, "launches": arr[i].launches,
                 "flops_per_tile": arr[i].flops_per_tile} for i in range(n.value)]


def crop_preprocess(crop_u8, mean, std, out_h, out_w, out=None):
    """uint8 BGR crop [h,w,3] on the GPU -> normalised, bilinearly resized fp32 [3,out_h,out_w]
    (VisualizeResults_iou.py:107-116 for a crop that is not network-sized)."""
    lib = _lib.load()
    crop_u8 = crop_u8.contiguous()
    h, w, _ = crop_u8.shape
    if out is None:
        out = torch.empty((3, out_h, out_w), dtype=torch.float32, device=crop_u8.device)
    with torch.cuda.device(crop_u8.device):
        _lib.check(lib.gs_crop_preprocess(crop_u8.data_ptr(), h, w, _lib.fptr3(mean), _lib.fptr3(std), out_h, out_w,
                                          out.data_ptr(), _stream_ptr(crop_u8.device)))
    return out


def mask_resize_nearest(mask, out_h, out_w):
    """uint8 class map [h,w] on the GPU -> [out_h,out_w], cv2.INTER_NEAREST sampling (:129)."""
    lib = _lib.load()
    mask = mask.contiguous()
    h, w = mask.shape
    out = torch.empty((out_h, out_w), dtype=torch.uint8, device=mask.device)
    with torch.cuda.device(mask.device):
        _lib.check(lib.gs_mask_resize_nearest(mask.data_ptr(), h, w, out_h, out_w, out.data_ptr(), _stream_ptr(mask.device)))
    return out


def paste_target(slide_map, ds=8, luts=None):
    """_lib.PasteTarget for a uint8 [map_h,map_w] GPU tensor (the 1/ds slide map of composite.SlideCompositor)."""
    if slide_map.dtype != torch.uint8 or slide_map.dim() != 2 or not slide_map.is_cuda or not slide_map.is_contiguous():
        raise ValueError("slide_map must be a contiguous uint8 [map_h,map_w] tensor on the GPU")
    t = _lib.PasteTarget()
    t.slide_map = slide_map.data_ptr()
    t.map_h, t.map_w, t.ds = int(slide_map.shape[0]), int(slide_map.shape[1]), int(ds)
    t.sx_lut = luts[0].data_ptr() if luts is not None else None
    t.sy_lut = luts[1].data_ptr() if luts is not None else None
    return t


def segment_crops_host(engines, mean_stds, crops, net_h=512, net_w=1024, batch=32, want_masks=True, want_net_maps=False,
                       want_hist=True, paste=None, origins=None, overlay=None):
    """gs_espnet_segment_crops_host: crops of any sizes in host memory -> per-crop class maps at crop size.

    engines: one EspnetEngine (the plain model) or several (the cfg-5 ensemble, each with its own (mean, std) in mean_stds).
    crops: list of uint8 BGR [h,w,3] numpy arrays / CPU tensors (pinned ones are DMA'd in place).
    paste: an _lib.PasteTarget (see paste_target) + origins [(x1, y1), ...]: the crops are also max-composited into the
    slide map on the GPU.
    overlay: (palette uint8 [k,3] RGB, wa, wb): also return every crop's class map coloured and blended over the crop,
    cv2.addWeighted(crop, wa, colour, wb, 0) (VisualizeResults_iou.py:139-146), computed on the GPU for the whole batch.
    Returns dict(masks=list of uint8 [h,w] numpy views of one pinned buffer | None, net_maps=uint8 [n,net_h,net_w] | None,
    counts=int64 [n,classes] counts of the crop-size maps | None, overlays=list of uint8 [h,w,3] BGR views | None).
    """
    lib = _lib.load()
    n = len(crops)
    if n == 0:
        return {"masks": [] if want_masks else None, "net_maps": None, "counts": None, "overlays": [] if overlay is not None else None}
    keep = []      # keeps converted inputs alive for the duration of the call
    ptrs = (ctypes.c_void_p * n)()
    hs, ws = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    for i, c in enumerate(crops):
        if isinstance(c, torch.Tensor):
            if c.is_cuda or c.dtype != torch.uint8 or c.dim() != 3 or c.shape[2] != 3 or not c.is_contiguous():
                raise ValueError("crop %d: expected a contiguous uint8 [h,w,3] CPU tensor" % i)
            ptrs[i] = c.data_ptr()
        else:
            c = np.ascontiguousarray(c, dtype=np.uint8)
            if c.ndim != 3 or c.shape[2] != 3:
                raise ValueError("crop %d: expected uint8 [h,w,3]" % i)
            ptrs[i] = c.ctypes.data
        keep.append(c)
        hs[i], ws[i] = int(c.shape[0]), int(c.shape[1])
    eng0 = engines[0]
    handles = (ctypes.c_void_p * len(engines))(*[e.handle for e in engines])
    means = (ctypes.c_float * (3 * len(engines)))(*[float(v) for ms in mean_stds for v in ms[0]])
    stds = (ctypes.c_float * (3 * len(engines)))(*[float(v) for ms in mean_stds for v in ms[1]])
    out_ptrs, out_buf, offs = None, None, None
    if want_masks:       # one pinned buffer for all maps: the library DMAs every map straight to its place
        sizes = [int(hs[i]) * int(ws[i]) for i in range(n)]
        offs = np.zeros(n + 1, dtype=np.int64)
        # 256-byte slots, the alignment of the library's packed device buffer: a batch's maps then sit exactly as they do on the
        # device and leave it as one DMA instead of one per crop (gs_espnet_segment_crops_host)
        np.cumsum([(s + 255) // 256 * 256 for s in sizes], out=offs[1:])
        out_buf = torch.empty(int(offs[-1]), dtype=torch.uint8, pin_memory=True)
        base = out_buf.data_ptr()
        out_ptrs = (ctypes.c_void_p * n)(*[base + int(offs[i]) for i in range(n)])
    net = torch.empty((n, net_h, net_w), dtype=torch.uint8, pin_memory=True) if want_net_maps else None
    hist = torch.zeros((n, eng0.classes), dtype=torch.int64, pin_memory=True) if want_hist else None
    x1 = y1 = None
    if paste is not None:
        if origins is None or len(origins) != n:
            raise ValueError("paste needs one (x1, y1) level-0 origin per crop")
        x1 = (ctypes.c_int * n)(*[int(o[0]) for o in origins])
        y1 = (ctypes.c_int * n)(*[int(o[1]) for o in origins])
    ov, ov_buf, ov_offs, pal = None, None, None, None
    if overlay is not None:     # one pinned buffer laid out like the library's packed input: a batch's overlays leave in one DMA
        palette, wa, wb = overlay
        pal = np.ascontiguousarray(palette, dtype=np.uint8).reshape(-1, 3)
        ov_offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([(int(hs[i]) * int(ws[i]) * 3 + 255) // 256 * 256 for i in range(n)], out=ov_offs[1:])
        ov_buf = torch.empty(int(ov_offs[-1]), dtype=torch.uint8, pin_memory=True)
        ov = _lib.CropOverlay()
        ov.palette_rgb = pal.ctypes.data
        ov.n_colours = int(pal.shape[0])
        ov.wa, ov.wb = float(wa), float(wb)
        ov_ptrs = (ctypes.c_void_p * n)(*[ov_buf.data_ptr() + int(ov_offs[i]) for i in range(n)])
        ov.out_bgr = ctypes.cast(ov_ptrs, ctypes.POINTER(ctypes.c_void_p))
        keep.append(ov_ptrs)
    for e in engines:
        e.quiesce()
    with torch.cuda.device(eng0.device):
        _lib.check(lib.gs_espnet_segment_crops_host(
            handles, len(engines), ptrs, hs, ws, n, means, stds, net_h, net_w, batch, out_ptrs,
            ctypes.c_void_p(net.data_ptr()) if net is not None else None,
            ctypes.c_void_p(hist.data_ptr()) if hist is not None else None,
            ctypes.byref(paste) if paste is not None else None, x1, y1, ctypes.byref(ov) if ov is not None else None))
    masks = None
    if want_masks:
        flat = out_buf.numpy()
        masks = [flat[int(offs[i]):int(offs[i]) + int(hs[i]) * int(ws[i])].reshape(int(hs[i]), int(ws[i])) for i in range(n)]
    overlays = None
    if ov is not None:
        flat = ov_buf.numpy()
        overlays = [flat[int(ov_offs[i]):int(ov_offs[i]) + int(hs[i]) * int(ws[i]) * 3].reshape(int(hs[i]), int(ws[i]), 3) for i in range(n)]
    return {"masks": masks, "net_maps": net.numpy() if net is not None else None,
            "counts": hist.numpy() if hist is not None else None, "overlays": overlays}


def ensemble_segment(engines, tiles_u8, mean_stds):
    """cfg 5: mean over models of softmax(logits_k) (each model with its own mean/std) -> argmax mask,
    per-class counts.  The reference has no ensemble code; the definition is this build's."""
    lib = _lib.load()
    tiles_u8 = tiles_u8.contiguous()
    n, h, w, _ = tiles_u8.shape
    dev = tiles_u8.device
    handles = (ctypes.c_void_p * len(engines))(*[e.handle for e in engines])
    means = (ctypes.c_float * (3 * len(engines)))(*[float(v) for ms in mean_stds for v in ms[0]])
    stds = (ctypes.c_float * (3 * len(engines)))(*[float(v) for ms in mean_stds for v in ms[1]])
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    hist = torch.empty((n, engines[0].classes), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gs_espnet_ensemble_forward(handles, len(engines), tiles_u8.data_ptr(), n, h, w, means, stds,
                                                  mask.data_ptr(), hist.data_ptr(), _stream_ptr(dev)))
    return mask, hist
