"""TEST INFRASTRUCTURE -- the graph of csrc/detector.hip restated over torch CPU operators and numpy.
Only tests/ may import this module.  NOT reference parity: the reference's detector is an external TensorFlow frozen
graph (module/faster-rcnn/detect_glomus_test.py:419-427) that is not in the reference; this checks that the GPU
assembly computes the graph its header describes (TF object-detection Faster R-CNN conventions: 2/255 x - 1
preprocessing, grid anchors, faster_rcnn_box_coder with scale factors 10/10/5/5, tf.image.crop_and_resize,
greedy non_max_suppression, zero-padded proposals and detections).
"""
import numpy as np
import torch
import torch.nn.functional as F

A = 12
STRIDE = 16
BASE = 256.0
PRE_NMS = 1024
PROPOSALS = 300
CROP = 14
MAX_DET = 100
LIM = np.float32(4.135166556742356)


def _conv(x, sd, name, stride, pad, relu):
    """x: [N,C,H,W] torch; weights stored [k,k,cin,cout]"""
    w = torch.from_numpy(sd[name + ".weight"]).permute(3, 2, 0, 1).contiguous()
    y = F.conv2d(x, w, torch.from_numpy(sd[name + ".bias"]), stride=stride, padding=pad)
    return F.relu(y) if relu else y


def preprocess(images_u8):
    """[N,H,W,3] uint8 -> [N,16,ceil(H/2),ceil(W/2)] fp32: 2/255 x - 1, 2x2 space-to-depth (channel (dy*2+dx)*3+c), 4 zero channels"""
    n, h, w, _ = images_u8.shape
    h2, w2 = (h + 1) // 2, (w + 1) // 2
    x = np.zeros((n, 2 * h2, 2 * w2, 3), dtype=np.float32)
    x[:, :h, :w] = images_u8.astype(np.float32) * np.float32(2.0 / 255.0) - np.float32(1.0)
    out = np.zeros((n, 16, h2, w2), dtype=np.float32)
    for dy in range(2):
        for dx in range(2):
            for c in range(3):
                out[:, (dy * 2 + dx) * 3 + c] = x[:, dy::2, dx::2, c]
    return torch.from_numpy(out)


def backbone(images_u8, sd):
    x = preprocess(images_u8)
    x = _conv(x, sd, "backbone.c1", 1, 1, True)
    x = F.max_pool2d(x, 3, 2, 1)
    x = _conv(x, sd, "backbone.c2", 1, 1, True)
    x = _conv(x, sd, "backbone.c3", 2, 1, True)
    x = _conv(x, sd, "backbone.c4", 1, 1, True)
    x = _conv(x, sd, "backbone.c5", 2, 1, True)
    return _conv(x, sd, "backbone.c6", 1, 1, True)          # [N,256,hf,wf]


def rpn(features, sd):
    r = _conv(features, sd, "rpn.conv", 1, 1, True)
    return _conv(r, sd, "rpn.head", 1, 0, False)             # [N,72,hf,wf]


def anchors(hf, wf):
    """[hf*wf*A, 4] (ymin,xmin,ymax,xmax) px; anchor a = scale index a//3, ratio index a%3, centre (cy*16, cx*16)"""
    scales = np.array([0.25, 0.5, 1.0, 2.0], dtype=np.float32)
    rs = np.array([0.70710678118654752, 1.0, 1.41421356237309505], dtype=np.float32)
    out = np.empty((hf, wf, A, 4), dtype=np.float32)
    for a in range(A):
        ah = scales[a // 3] / rs[a % 3] * np.float32(BASE)
        aw = scales[a // 3] * rs[a % 3] * np.float32(BASE)
        yc = (np.arange(hf, dtype=np.float32) * STRIDE)[:, None]
        xc = (np.arange(wf, dtype=np.float32) * STRIDE)[None, :]
        out[:, :, a, 0] = yc - np.float32(0.5) * ah
        out[:, :, a, 1] = xc - np.float32(0.5) * aw
        out[:, :, a, 2] = yc + np.float32(0.5) * ah
        out[:, :, a, 3] = xc + np.float32(0.5) * aw
    return out.reshape(-1, 4)


def decode_clip(anc, d, H, W):
    """faster_rcnn_box_coder decode (scale factors 10,10,5,5), size deltas bounded at log(1000/16), clip to the window"""
    f = np.float32
    ha, wa = anc[:, 2] - anc[:, 0], anc[:, 3] - anc[:, 1]
    yca, xca = anc[:, 0] + f(0.5) * ha, anc[:, 1] + f(0.5) * wa
    th = np.minimum(d[:, 2] * f(0.2), LIM)
    tw = np.minimum(d[:, 3] * f(0.2), LIM)
    hh, ww = np.exp(th).astype(f) * ha, np.exp(tw).astype(f) * wa
    yc, xc = d[:, 0] * f(0.1) * ha + yca, d[:, 1] * f(0.1) * wa + xca
    out = np.stack([yc - f(0.5) * hh, xc - f(0.5) * ww, yc + f(0.5) * hh, xc + f(0.5) * ww], 1)
    out[:, 0::2] = np.clip(out[:, 0::2], 0, f(H))
    out[:, 1::2] = np.clip(out[:, 1::2], 0, f(W))
    return out.astype(f)


def iou(a, b):
    aa = (a[2] - a[0]) * (a[3] - a[1])
    ab = (b[2] - b[0]) * (b[3] - b[1])
    if aa <= 0 or ab <= 0:
        return np.float32(0)
    ih = max(min(a[2], b[2]) - max(a[0], b[0]), np.float32(0))
    iw = max(min(a[3], b[3]) - max(a[1], b[1]), np.float32(0))
    inter = np.float32(ih * iw)
    return np.float32(inter / np.float32(np.float32(aa + ab) - inter))


def nms_sorted(boxes, scores, thr, score_thr, max_out):
    """greedy NMS over score-descending candidates -> kept positions"""
    keep = []
    dead = np.zeros(len(boxes), dtype=bool)
    for i in range(len(boxes)):
        if len(keep) >= max_out or not scores[i] > score_thr:
            break
        if dead[i]:
            continue
        keep.append(i)
        for j in range(i + 1, len(boxes)):
            if not dead[j] and iou(boxes[i], boxes[j]) > thr:
                dead[j] = True
    return keep


def top_k(scores, k):
    order = np.argsort(-scores.astype(np.float64), kind="stable")[:k]      # ties: lower index first
    return order, scores[order]


def proposals_from_rpn(rpn_nhwc, H, W, rpn_iou=0.7):
    """rpn_nhwc: [hf,wf,72] of ONE image -> (proposals [300,4] px zero padded, valid count)"""
    hf, wf, _ = rpn_nhwc.shape
    cls = rpn_nhwc[:, :, :2 * A].reshape(-1, A, 2)
    box = rpn_nhwc[:, :, 2 * A:].reshape(-1, A, 4)
    score = (np.float32(1) / (np.float32(1) + np.exp((cls[:, :, 0] - cls[:, :, 1]).astype(np.float32)))).astype(np.float32).reshape(-1)
    idx, sc = top_k(score, PRE_NMS)
    anc = anchors(hf, wf)[idx]
    boxes = decode_clip(anc, box.reshape(-1, 4)[idx], H, W)
    keep = nms_sorted(boxes, sc, np.float32(rpn_iou), np.float32(0), PROPOSALS)
    out = np.zeros((PROPOSALS, 4), dtype=np.float32)
    out[:len(keep)] = boxes[keep]
    return out, len(keep)


def crop_and_resize(feat_hwc, boxes_norm, crop):
    """tf.image.crop_and_resize, bilinear, extrapolation 0, for ONE image: [h,w,c], [k,4] -> [k,crop,crop,c]"""
    h, w, c = feat_hwc.shape
    f = np.float32
    out = np.zeros((len(boxes_norm), crop, crop, c), dtype=f)
    for b, (y1, x1, y2, x2) in enumerate(boxes_norm.astype(f)):
        hs = (y2 - y1) * f(h - 1) / f(crop - 1)
        ws = (x2 - x1) * f(w - 1) / f(crop - 1)
        for y in range(crop):
            in_y = y1 * f(h - 1) + f(y) * hs
            if in_y < 0 or in_y > h - 1:
                continue
            ty, by = int(np.floor(in_y)), int(np.ceil(in_y))
            fy = f(in_y - f(ty))
            for x in range(crop):
                in_x = x1 * f(w - 1) + f(x) * ws
                if in_x < 0 or in_x > w - 1:
                    continue
                lx, rx = int(np.floor(in_x)), int(np.ceil(in_x))
                fx = f(in_x - f(lx))
                top = feat_hwc[ty, lx] + (feat_hwc[ty, rx] - feat_hwc[ty, lx]) * fx
                bot = feat_hwc[by, lx] + (feat_hwc[by, rx] - feat_hwc[by, lx]) * fx
                out[b, y, x] = top + (bot - top) * fy
    return out


def box_head(features_hwc, proposals, H, W, sd):
    """ONE image: features [hf,wf,256], proposals [300,4] px -> head outputs [300,6]"""
    norm = proposals / np.array([H, W, H, W], dtype=np.float32)
    crops = crop_and_resize(features_hwc, norm.astype(np.float32), CROP)               # [300,14,14,256]
    x = torch.from_numpy(crops).permute(0, 3, 1, 2)
    x = F.max_pool2d(x, 2, 2)
    x = _conv(x, sd, "head.h1", 1, 0, True)
    x = _conv(x, sd, "head.h2", 2, 1, True)                                            # [300,128,4,4]
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], 16, -1)
    pooled = torch.zeros(x.shape[0], x.shape[2])
    for i in range(16):                                                                # sum in index order, then / 16
        pooled = pooled + x[:, i]
    pooled = pooled / 16
    return _conv(pooled[:, :, None, None], sd, "head.fc", 1, 0, False)[:, :, 0, 0].numpy()


def detections_from_head(head, proposals, n_valid, H, W, det_iou=0.6, score_thr=0.0):
    """-> (boxes [100,4] normalised, scores [100], classes [100], num)"""
    f = np.float32
    score = (f(1) / (f(1) + np.exp((head[:, 0] - head[:, 1]).astype(f)))).astype(f)
    score[n_valid:] = -1
    boxes = decode_clip(proposals, head[:, 2:], H, W)
    idx, sc = top_k(score, len(score))
    keep = nms_sorted(boxes[idx], sc, f(det_iou), f(score_thr), MAX_DET)
    ob = np.zeros((MAX_DET, 4), dtype=f)
    os_ = np.zeros(MAX_DET, dtype=f)
    oc = np.zeros(MAX_DET, dtype=f)
    ob[:len(keep)] = boxes[idx][keep] / np.array([H, W, H, W], dtype=f)
    os_[:len(keep)] = sc[keep]
    oc[:len(keep)] = 1
    return ob, os_, oc, len(keep)


def detect(images_u8, sd):
    """the whole graph for a small batch -> (boxes [N,100,4], scores, classes, num) + intermediates"""
    n, H, W, _ = images_u8.shape
    with torch.no_grad():
        feats = backbone(images_u8, sd)
        r = rpn(feats, sd)
    feats = feats.permute(0, 2, 3, 1).contiguous().numpy()
    r = r.permute(0, 2, 3, 1).contiguous().numpy()
    res = {"features": feats, "rpn": r, "proposals": [], "head": [], "boxes": [], "scores": [], "classes": [], "num": []}
    for i in range(n):
        prop, nv = proposals_from_rpn(r[i], H, W)
        with torch.no_grad():
            head = box_head(feats[i], prop, H, W, sd)
        b, s, c, k = detections_from_head(head, prop, nv, H, W)
        for key, v in (("proposals", prop), ("head", head), ("boxes", b), ("scores", s), ("classes", c), ("num", k)):
            res[key].append(v)
    return {k: np.asarray(v) for k, v in res.items()}
