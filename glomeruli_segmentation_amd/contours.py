"""Class map -> polygons -> labelme-style JSON: the per-crop export of
module/espnet/test/VisualizeResults_iou.py:161-182 over module/common/boundary_extractor.py.

The contour tracer, the arc length and the polygon simplifier are host functions of libglomseg.so (csrc/contours.cpp)
restated from the published algorithms of OpenCV 4.3 (cv2 is not installed).  They are checked point for point -- contour
count, list order, start point, direction, every point and every polygon vertex -- against oracle/contour_oracle.py, a
second restatement written independently of the C++ (tests/test_contour_oracle.py); cv2's own output cannot be produced
on this stack.
"""
import ctypes
import json

import numpy as np

from . import _lib

LABEL_IDX = {1: "glomerulus", 2: "crescent", 3: "sclerosis", 4: "mesangium"}     # VisualizeResults_iou.py:47-52


def find_contours(binary, simple=True):
    """cv2.findContours(img, RETR_LIST, CHAIN_APPROX_SIMPLE)[0]: list of int32 [n,1,2] arrays (x, y)."""
    lib = _lib.load()
    img = np.ascontiguousarray(binary, dtype=np.uint8)
    h, w = img.shape
    nc, npts = ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.gs_find_contours(img.ctypes.data_as(ctypes.c_void_p), h, w, int(simple), None, 0, None, 0,
                                    ctypes.byref(nc), ctypes.byref(npts)))
    pts = np.empty((max(npts.value, 1), 2), dtype=np.int32)
    off = np.empty(nc.value + 1, dtype=np.int32)
    _lib.check(lib.gs_find_contours(img.ctypes.data_as(ctypes.c_void_p), h, w, int(simple),
                                    pts.ctypes.data_as(ctypes.c_void_p), pts.shape[0], off.ctypes.data_as(ctypes.c_void_p),
                                    off.shape[0], ctypes.byref(nc), ctypes.byref(npts)))
    return [pts[off[k]:off[k + 1]].reshape(-1, 1, 2).copy() for k in range(nc.value)]


def arc_length(contour):
    """cv2.arcLength(contour, True)"""
    xy = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
    return float(_lib.load().gs_arc_length_closed(xy.ctypes.data_as(ctypes.c_void_p), len(xy)))


def approx_poly(contour, epsilon):
    """cv2.approxPolyDP(contour, epsilon, True): int32 [m,1,2]"""
    xy = np.ascontiguousarray(contour, dtype=np.int32).reshape(-1, 2)
    out = np.empty_like(xy)
    m = _lib.load().gs_approx_poly_closed(xy.ctypes.data_as(ctypes.c_void_p), len(xy), float(epsilon),
                                          out.ctypes.data_as(ctypes.c_void_p))
    return out[:m].reshape(-1, 1, 2)


def bound2line(class_map, max_classes=-1, g_min_point=200, o_min_points=50, g_epsilon=0.003, o_epsilon=0.002):
    """boundary_extractor.bound2line (:6-50): {class: [polygon int32 [m,2], ...]}.  Class 1 is traced on
    `class_map >= 1` (the whole glomerulus), the others on equality; contours with fewer than the minimum
    number of (simplified-chain) points are noise."""
    class_map = np.asarray(class_map)
    num_class = int(class_map.max()) + 1 if max_classes < 0 else min(max_classes, int(class_map.max()) + 1)
    out = {}
    for cls in range(1, num_class):
        mask = (class_map >= cls) if cls == 1 else (class_map == cls)
        min_points, eps = (g_min_point, g_epsilon) if cls == 1 else (o_min_points, o_epsilon)
        conts = [c for c in find_contours(mask.astype(np.uint8)) if len(c) >= min_points]
        if conts:
            out[cls] = [np.squeeze(approx_poly(c, eps * arc_length(c))) for c in conts]
    return out


def labelme_dict(class_map, image_name, class_map_path=None):
    """the JSON body of VisualizeResults_iou.py:161-177 (imageData is left to the caller: the reference
    stores the original crop there)."""
    lines = bound2line(class_map, max_classes=4)
    shapes = []
    for idx, label in LABEL_IDX.items():
        for poly in lines.get(idx, []):
            shapes.append({"line_color": None, "points": np.asarray(poly).reshape(-1, 2).tolist(), "fill_color": None,
                           "label": label})
    d = {"shapes": shapes, "lineColor": [0, 0, 0, 255], "imagePath": image_name, "flags": {}, "fillColor": [0, 0, 0, 255]}
    if class_map_path:
        d["classMapPath"] = class_map_path
    return d
