"""SURVEY 8f-4, the polygon / JSON export (module/common/boundary_extractor.py:33-47): the product's host functions
(csrc/contours.cpp through glomeruli_segmentation_amd.contours) against oracle/contour_oracle.py, a second restatement of
OpenCV 4.3's findContours(RETR_LIST, CHAIN_APPROX_SIMPLE) / arcLength / approxPolyDP written independently from the
published algorithm.  Equality is exact: contour count, list order, start point, direction, every point, every polygon
vertex -- these decide `len(x) >= min_points` (:42) and so whether a shape appears in the JSON at all."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from glomeruli_segmentation_amd import contours as prod
from oracle import contour_oracle as orc


def same_contours(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and x.dtype == y.dtype and (x == y).all() for x, y in zip(a, b))


def same_lines(a, b):
    """bound2line dict equality: same classes, same polygons in the same order, vertex for vertex"""
    if sorted(a) != sorted(b):
        return False
    for k in a:
        if len(a[k]) != len(b[k]):
            return False
        for p, q in zip(a[k], b[k]):
            p, q = np.asarray(p), np.asarray(q)
            if p.shape != q.shape or (p != q).any():
                return False
    return True


def check_image(img):
    for simple, method in ((True, orc.CHAIN_APPROX_SIMPLE), (False, orc.CHAIN_APPROX_NONE)):
        a, b = prod.find_contours(img, simple=simple), orc.find_contours(img, method)
        assert same_contours(a, b), (simple, len(a), len(b))
    for c in b if len(b) < 40 else b[:40]:
        assert prod.arc_length(c) == orc.arc_length(c)
    for c in orc.find_contours(img, orc.CHAIN_APPROX_SIMPLE)[:40]:
        arc = orc.arc_length(c)
        assert prod.arc_length(c) == arc                       # bit for bit: float32 segments, double sum
        for k in (0.003, 0.002, 0.02, 0.0):                    # the reference's two factors (:7-8), a coarse one, and zero
            pa, pb = prod.approx_poly(c, k * arc), orc.approx_poly_dp(c, k * arc)
            assert pa.shape == pb.shape and (pa == pb).all(), (k, len(c), pa[:, 0].tolist(), pb[:, 0].tolist())


def blob_class_map(rng, h, w):
    """a glomerulus-like class map: a big class-1 region with islands of classes 2..4 and background holes"""
    sigma = rng.uniform(2.0, 9.0)
    body = ndi.gaussian_filter(rng.standard_normal((h, w)), sigma) > rng.normal(0, 0.01)
    cm = body.astype(np.uint8)
    for cls in (2, 3, 4):
        isl = ndi.gaussian_filter(rng.standard_normal((h, w)), rng.uniform(1.0, 5.0)) > rng.uniform(0.02, 0.08)
        cm[body & isl] = cls
    if rng.random() < 0.5:                                      # speckle: one- and two-pixel domains, touching corners
        sp = rng.random((h, w)) < 0.01
        cm[sp] = rng.integers(0, 5, size=int(sp.sum()))
    return cm


def test_oracle_known_answers():
    """what OpenCV is known to return for the textbook cases -- pins the oracle itself before it is used as one"""
    img = np.zeros((40, 60), np.uint8)
    img[5:25, 10:40] = 255
    img[10:20, 20:30] = 0
    img[30, 50] = 255
    cs = orc.find_contours(img)
    # RETR_LIST: last found first.  Outer borders run counter-clockwise on the screen from the top-left pixel (down first);
    # hole borders run clockwise over the foreground ring, from the pixel left of the hole's first pixel, with cut corners.
    assert [c[:, 0].tolist() for c in cs] == [
        [[50, 30]],
        [[19, 10], [20, 9], [29, 9], [30, 10], [30, 19], [29, 20], [20, 20], [19, 19]],
        [[10, 5], [10, 24], [39, 24], [39, 5]]]
    assert orc.arc_length(cs[2]) == 2 * (29 + 19)
    assert abs(orc.arc_length(cs[1]) - (4 * 9 + 4 * 2 ** 0.5)) < 1e-6
    assert orc.arc_length(cs[0]) == 0.0
    # a one-pixel-wide horizontal line: out and back, two points; vertical likewise; a diagonal one too
    line = np.zeros((5, 12), np.uint8)
    line[2, 3:9] = 1
    assert orc.find_contours(line)[0][:, 0].tolist() == [[3, 2], [8, 2]]
    assert len(orc.find_contours(line, orc.CHAIN_APPROX_NONE)[0]) == 10          # 6 pixels out, 4 back
    diag = np.eye(6, dtype=np.uint8)
    assert orc.find_contours(diag)[0][:, 0].tolist() == [[0, 0], [5, 5]]
    # approxPolyDP: the hand-worked case of tests/test_host_logic.py (starts at the cut, not at the first input point)
    rect = np.array([[5, 0], [10, 0], [10, 2], [5, 2], [0, 2], [0, 0]], dtype=np.int32)
    assert orc.approx_poly_dp(rect, 0.5)[:, 0].tolist() == [[0, 0], [10, 0], [10, 2], [0, 2]]
    assert orc.approx_poly_dp(rect, 100.0)[:, 0].tolist() == [[0, 0]]
    # open curve (not on the reference's path; the restatement keeps it): end points stay, the middle goes
    poly = np.array([[0, 0], [5, 1], [10, 0]], dtype=np.int32)
    assert orc.approx_poly_dp(poly, 2.0, closed=False)[:, 0].tolist() == [[0, 0], [10, 0]]
    assert orc.approx_poly_dp(poly, 0.5, closed=False)[:, 0].tolist() == [[0, 0], [5, 1], [10, 0]]


def test_oracle_against_scipy_topology():
    """the oracle itself against an implementation this repo did not write: scipy.ndimage says how many 8-connected components
    and enclosed 4-connected background components an image has (= outer borders + hole borders) and which foreground pixels
    touch background by a 4-neighbour (= the points CHAIN_APPROX_NONE visits); every traced border is a closed 8-connected
    chain, and CHAIN_APPROX_SIMPLE keeps exactly the chain's direction changes"""
    rng = np.random.default_rng(17)
    s8, s4 = np.ones((3, 3), int), ndi.generate_binary_structure(2, 1)
    for trial in range(10):
        field = ndi.gaussian_filter(rng.standard_normal((90, 130)), 0.8 + trial)
        img = (field > 0.03 / (1 + trial)).astype(np.uint8)
        full = orc.find_contours(img, orc.CHAIN_APPROX_NONE)
        simple = orc.find_contours(img, orc.CHAIN_APPROX_SIMPLE)
        _, ncomp = ndi.label(img, structure=s8)
        bl, nb = ndi.label(1 - img, structure=s4)
        frame = (set(bl[0, :]) | set(bl[-1, :]) | set(bl[:, 0]) | set(bl[:, -1])) - {0}
        assert len(full) == len(simple) == ncomp + (nb - len(frame)), trial
        traced = set()
        for c, cs in zip(full, simple):
            pts = c[:, 0, :]
            traced |= set(map(tuple, pts.tolist()))
            if len(pts) > 1:
                step = np.abs(np.diff(np.vstack([pts, pts[:1]]), axis=0)).max(axis=1)
                assert (step == 1).all()                                   # a closed 8-connected chain
            # SIMPLE = the points where the chain's direction changes (cyclically), in the same order
            d = np.diff(np.vstack([pts[-1:], pts, pts[:1]]), axis=0)
            turn = [tuple(p) for p, a, b in zip(pts.tolist(), d[:-1].tolist(), d[1:].tolist()) if a != b] if len(pts) > 1 else [tuple(pts[0])]
            assert [tuple(p) for p in cs[:, 0, :].tolist()] == turn, trial
        pad = np.pad(img, 1)
        border = (img == 1) & ((pad[:-2, 1:-1] == 0) | (pad[2:, 1:-1] == 0) | (pad[1:-1, :-2] == 0) | (pad[1:-1, 2:] == 0))
        assert traced == set(zip(*np.nonzero(border)[::-1])), trial


def test_hole_start_on_a_diagonal_is_not_a_vertex():
    """the case that separated the product from OpenCV until round 4: a hole whose scan start pixel lies inside a straight
    SW-NE run of its border -- icvFetchContour starts with prev_s = s ^ 4, so that pixel is not written"""
    img = np.zeros((9, 9), np.uint8)
    img[1:8, 1:8] = 1
    for k in range(4):                       # a triangular hole whose left wall is the anti-diagonal
        img[2 + k, 5 - k:6] = 0
    cs = orc.find_contours(img)
    hole = cs[0][:, 0].tolist()
    full = orc.find_contours(img, orc.CHAIN_APPROX_NONE)[0][:, 0].tolist()
    assert full[0] == [4, 2] and full[0] not in hole         # the start pixel of the trace is not a SIMPLE vertex
    assert same_contours(prod.find_contours(img), cs)


def test_random_noise_images():
    """dense random pixels: every degenerate neighbourhood (pixels visited 2-4 times, touching holes, 1-wide strokes)"""
    rng = np.random.default_rng(11)
    for trial in range(240):
        h, w = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        img = (rng.random((h, w)) < rng.choice([0.15, 0.35, 0.5, 0.65, 0.85])).astype(np.uint8) * 255
        check_image(img)


def test_degenerate_shapes():
    cases = []
    one = np.zeros((1, 1), np.uint8); one[0, 0] = 1
    cases.append(one)                                                   # the whole image is one pixel
    cases.append(np.ones((1, 7), np.uint8))                             # one row
    cases.append(np.ones((7, 1), np.uint8))                             # one column
    cases.append(np.ones((6, 9), np.uint8))                             # all foreground: the border touches every edge
    cases.append(np.zeros((6, 9), np.uint8))                            # nothing
    ring = np.ones((7, 7), np.uint8); ring[1:6, 1:6] = 0
    cases.append(ring)                                                  # 1-wide ring on the image border
    plus = np.zeros((9, 9), np.uint8); plus[4, :] = 1; plus[:, 4] = 1
    cases.append(plus)                                                  # 1-wide cross touching all four edges
    tee = np.zeros((9, 11), np.uint8); tee[2, 1:10] = 1; tee[2:8, 5] = 1
    cases.append(tee)
    chk = (np.indices((10, 10)).sum(0) % 2).astype(np.uint8)
    cases.append(chk)                                                   # checkerboard: one 8-connected net, many holes
    cases.append(1 - chk)
    two = np.ones((7, 9), np.uint8); two[1:3, 1:3] = 0; two[3:5, 3:5] = 0; two[5, 5] = 0
    cases.append(two)                                                   # holes that touch holes by their corners
    hh = np.ones((8, 8), np.uint8); hh[2, 2:6] = 0; hh[3:6, 2] = 0; hh[4, 4] = 0
    cases.append(hh)
    frame = np.zeros((12, 12), np.uint8); frame[0, :] = frame[-1, :] = 1; frame[:, 0] = frame[:, -1] = 1; frame[5:7, 5:7] = 1
    cases.append(frame)                                                 # an island inside a hole inside a border-touching frame
    zig = np.zeros((8, 20), np.uint8)
    for x in range(20):
        zig[3 + (x % 2), x] = 1
    cases.append(zig)                                                   # 1-wide zig-zag
    for c in cases:
        check_image(c)
        check_image(np.ascontiguousarray(c.T))
        check_image(np.ascontiguousarray(c[::-1]))
        check_image(np.ascontiguousarray(c[:, ::-1]))


def shape_with_points(target):
    """searches a small family of staircases for one whose single outer border has exactly `target` SIMPLE points"""
    for n in range(2, 140):
        base = np.zeros((2 * n + 6, 2 * n + 6), np.uint8)
        for k in range(n):
            base[2:4 + 2 * k, 2 + 2 * k:4 + 2 * k] = 1
        for cut in range(0, 4):
            img = base.copy()
            if cut >= 1:
                img[2, 2] = 0                     # cut the top-left corner: one vertex becomes two
            if cut >= 2:
                img[2, 3] = 0
            if cut >= 3:
                img[3, 2] = 0
            cs = orc.find_contours(img)
            if len(cs) == 1 and len(cs[0]) == target:
                return img
    raise AssertionError("no shape with %d points" % target)


@pytest.mark.parametrize("target,cls,kept", [(200, 1, True), (199, 1, False), (50, 2, True), (49, 2, False),
                                             (50, 3, True), (49, 4, False)])
def test_min_points_filter_flips(target, cls, kept):
    """contours of exactly min_points / min_points - 1 simple-chain points: the noise filter of :42 flips between them"""
    img = shape_with_points(target)
    cm = np.zeros_like(img)
    if cls == 1:
        cm[img > 0] = 1
    else:
        big = np.zeros((img.shape[0] + 8, img.shape[1] + 8), np.uint8)
        big[1:-1, 1:-1] = 1                     # the class-1 body (4 corners: itself below g_min_point, never reported)
        big[4:-4, 4:-4][img > 0] = cls
        cm = big
    a, b = prod.bound2line(cm), orc.bound2line(cm)
    assert same_lines(a, b)
    assert (cls in b) == kept
    if kept:
        assert len(b[cls]) == 1 and len(b[cls][0]) >= 3
    # the same through the JSON body the driver writes
    d = prod.labelme_dict(cm, "x.PNG")
    assert sum(1 for s in d["shapes"] if s["label"] == prod.LABEL_IDX[cls]) == (1 if kept else 0)


def test_bound2line_on_random_class_maps():
    """>= 50 random blob class maps: product dict == oracle dict, with the reference's thresholds and with small ones
    (so that many contours pass the filter and are simplified), with and without max_classes=4 (VisualizeResults_iou.py:161)"""
    rng = np.random.default_rng(2024)
    n_shapes = 0
    for trial in range(56):
        h, w = int(rng.integers(60, 260)), int(rng.integers(60, 320))
        cm = blob_class_map(rng, h, w)
        for kw in (dict(max_classes=4), dict(), dict(max_classes=4, g_min_point=12, o_min_points=5),
                   dict(g_min_point=1, o_min_points=1, g_epsilon=0.01, o_epsilon=0.0005)):
            a, b = prod.bound2line(cm, **kw), orc.bound2line(cm, **kw)
            assert same_lines(a, b), (trial, kw)
            n_shapes += sum(len(v) for v in b.values())
        if trial < 12:
            for cls in range(1, 5):
                check_image(((cm >= cls) if cls == 1 else (cm == cls)).astype(np.uint8) * 255)
    assert n_shapes > 2000


def test_city_format_quirk_goes_through_both():
    """--cityFormat relabels before bound2line(max_classes=4): class 1 is then `>= 1` = the whole image,
    one 4-point border, below g_min_point -- nothing is reported, by both"""
    from glomeruli_segmentation_amd import imageops
    rng = np.random.default_rng(5)
    cm = imageops.relabel_city(blob_class_map(rng, 120, 160))
    a, b = prod.bound2line(cm, max_classes=4), orc.bound2line(cm, max_classes=4)
    assert same_lines(a, b) and 1 not in b
