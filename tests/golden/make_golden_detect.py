#!/usr/bin/env python3
"""Golden vectors for the detector's host logic, produced by the REFERENCE's own class
(module/faster-rcnn/detect_glomus_test.py: GlomusDetector.calc_window_size, the strides of scan_region /
scan_region_from_image, write_detected_result).

That script imports `tensorflow` and `openslide` (not installed) at module level; neither is touched by the functions
exercised here, so empty placeholder modules are registered under those names for the import only, and the module's
`datetime` is frozen so that the CSV rows are reproducible.  `detect_box` is NOT recorded: its `WINDOW_X * xmin` depends on
the NumPy version's promotion rules (float64 on the reference's NumPy 1.x, float32 under the NumPy 2 of this container), so
it is covered by a known-answer test written for the reference's stack instead.  Only arrays / strings are written.

    python tests/golden/make_golden_detect.py        ->  tests/golden/detect.npz
"""
import datetime as real_datetime
import io
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GS_REFERENCE", "/root/reference")
for name in ("tensorflow", "openslide"):
    sys.modules.setdefault(name, types.ModuleType(name))                 # import-time placeholders only
sys.path.insert(0, os.path.join(REF, "module", "faster-rcnn"))
sys.path.insert(0, os.path.join(REF, "module", "espnet", "test"))        # glomus_handler
import detect_glomus_test as ref  # noqa: E402


class FrozenDatetime(real_datetime.datetime):
    @classmethod
    def today(cls):
        return cls(2020, 1, 2, 3, 4, 5)


ref.datetime = types.SimpleNamespace(datetime=FrozenDatetime)

# (width, height, mpp_x, mpp_y, downsample, window_um, overlap)
CASES = [
    (53248, 23040, 0.2277, 0.2277, 8.0, 2000, 0.1),        # the example slide with the example flags
    (53248, 23040, 0.2277, 0.2277, 8.0, None, None),        # defaults: 500 um, 0.5
    (40000, 40000, 0.2277, 0.2277, 8.0, 2000, 0.1),        # BASELINE cfg 4
    (98304, 61440, 0.4530, 0.4549, 4.0, 1000, 0.25),
    (12345, 6789, 0.5, 0.25, 2.0, 300, 0.0),
    (20000, 30000, 0.1213, 0.1213, 16.0, 750, 0.6),
]

out = {"cases": np.array([[c[0], c[1], c[2], c[3], c[4], -1 if c[5] is None else c[5], -1 if c[6] is None else c[6]] for c in CASES],
                         dtype=np.float64)}
tmp = tempfile.mkdtemp()
geo, rows = [], []
for k, (w, h, mx, my, ds, win, ov) in enumerate(CASES):
    d = ref.GlomusDetector("OPT_PAS", "none.txt", tmp + "/site/", os.path.join(tmp, "a", "b", "c%d" % k), "_GlomusList", win, ov, 0.6)
    d.org_slide_width, d.org_slide_height, d.mpp_x, d.mpp_y, d.slide_downsample = w, h, mx, my, ds
    wxo, wyo, xs, ys, wx, wy = d.calc_window_size()
    # strides as scan_region (:266-268) and scan_region_from_image (:218-219) form them
    s0x, s0y = int(wxo * (1.0 - d.OVERLAP_RATIO)), int(wyo * (1.0 - d.OVERLAP_RATIO))
    sIx, sIy = int(wx * (1.0 - d.OVERLAP_RATIO)), int(wy * (1.0 - d.OVERLAP_RATIO))
    geo.append([wxo, wyo, xs, ys, wx, wy, s0x, s0y, sIx, sIy])
    f = io.StringIO()
    f.flush = lambda: None
    bs = [[10 + k, 20, 300 + 7 * k, 411, np.float32(0.91)], [0, 0, wx, wy, np.float32(0.6)], [5, 6, 7, 8, 0.0]]
    stdout = sys.stdout
    sys.stdout = io.StringIO()
    try:
        d.write_detected_result(bs, 1, 2, s0x * 1, s0y * 2, f, "site_a", "H16-%04d" % k, "H16-%04d_PAS.ndpi" % k)
        d.write_detected_result(bs[:1], 0, 0, sIx * 3 * ds, sIy * 1 * ds, f, "site_a", "H16-%04d" % k, "H16-%04d_PAS.PNG" % k)   # :234
    finally:
        sys.stdout = stdout
    rows.append(f.getvalue())
out["geometry"] = np.array(geo, dtype=np.float64)
out["rows"] = np.array(rows)
out["types"] = np.array([ref.GlomusHandler.get_staining_type(t) for t in ("OPT_PAS", "OPT_PAM", "OPT_MT", "OPT_Azan", "OPT_HE", "x")])
np.savez_compressed(os.path.join(HERE, "detect.npz"), **out)
print(out["geometry"][0], rows[0])
