#!/usr/bin/env python3
"""Golden vectors for the detection-merge step, produced by the REFERENCE's own class
(module/faster-rcnn/merge_overlaped_glomus.py: MargeOverlapedGlomus.check_overlap_from_list).

That script imports `openslide` (not installed) at module level but the merge arithmetic never
touches it, so an empty placeholder module is registered under that name for the import only.
Only arrays are written.   python tests/golden/make_golden_merge.py
"""
import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GS_REFERENCE", "/root/reference")
sys.modules.setdefault("openslide", types.ModuleType("openslide"))      # import-time placeholder only
sys.path.insert(0, os.path.join(REF, "module", "faster-rcnn"))
sys.path.insert(0, os.path.join(REF, "module", "espnet", "test"))       # glomus_handler
import merge_overlaped_glomus as ref  # noqa: E402


def run_ref(dets, mpp, thr, conf=0.6):
    m = ref.MargeOverlapedGlomus("OPT_PAS", "", "", "", conf, "", thr)
    m.rect_list = []
    tmp = []
    for d in dets:
        if float(d[4]) >= conf:
            tmp.append([float(d[0]), float(d[1]), float(d[2]), float(d[3]), float(d[4]),
                        (float(d[2]) - float(d[0])) * (float(d[3]) - float(d[1])), 0.0])
    m.check_overlap_from_list(tmp, mpp, mpp)
    return np.array([r[:5] for r in m.rect_list], dtype=np.float64).reshape(-1, 5)


def main():
    out = {}
    rng = np.random.default_rng(11)
    cases = []
    # (a) the 28 annotated glomeruli of the example slide (ds-8 pixels -> level 0), seen through
    #     overlapping windows: each box jittered 1-3 times, as overlapping sliding windows produce
    xml = os.path.join(REF, "example", "data", "02_PAS", "PAS-001", "annotations", "OPT_PAS_PAS-001_pw40_ds8.xml")
    boxes = []
    for obj in ET.parse(xml).getroot().iter("object"):
        bb = obj.find("bndbox")
        boxes.append([int(bb.find(k).text) * 8 for k in ("xmin", "ymin", "xmax", "ymax")])
    out["example_boxes"] = np.array(boxes, dtype=np.int64)
    dets = []
    for b in boxes:
        for _ in range(int(rng.integers(1, 4))):
            j = rng.normal(0, 40, 4)
            dets.append([b[0] + j[0], b[1] + j[1], b[2] + j[2], b[3] + j[3], float(rng.uniform(0.3, 1.0))])
    cases.append((np.array(dets), 0.2277, 0.35))
    # (b) dense random clutter at several scales / thresholds / mpp
    for k in range(12):
        n = int(rng.integers(5, 160))
        c = rng.uniform(0, 20000, (n, 2))
        s = rng.uniform(200, 2600, (n, 2))
        conf = rng.uniform(0.2, 1.0, n)
        d = np.concatenate([c, c + s, conf[:, None]], 1)
        cases.append((d, float(rng.choice([0.2277, 0.25, 0.5])), float(rng.choice([0.2, 0.35, 0.5, 0.8]))))
    # (c) integer-grid boxes: exact ties in area and overlap exercise the stable sorts
    for k in range(6):
        n = int(rng.integers(10, 80))
        c = rng.integers(0, 40, (n, 2)) * 250.0
        s = rng.integers(2, 8, (n, 2)) * 250.0
        conf = rng.integers(5, 10, n) / 10.0
        cases.append((np.concatenate([c, c + s, conf[:, None]], 1), 0.2277, 0.35))
    for i, (d, mpp, thr) in enumerate(cases):
        out["in_%d" % i] = d
        out["par_%d" % i] = np.array([mpp, thr])
        out["out_%d" % i] = run_ref(d, mpp, thr)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "merge.npz"), **out)
    print("merge.npz: %d cases, %d -> %d boxes in case 0" % (len(cases), len(cases[0][0]), len(out["out_0"])))


if __name__ == "__main__":
    main()
