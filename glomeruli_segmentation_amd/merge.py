"""Merging of overlapping glomerulus detections and the CSV / file-name contracts between the
detector, the merge step and the crop step.

Restates module/faster-rcnn/merge_overlaped_glomus.py (greedy insertion of detections, largest
area first, with the reference's merge rules and their quirks) and the row formats of
detect_glomus_test.py:319-325 (detections), merge_overlaped_glomus.py:121-124 (merged list) and
make_seg_data.py:248-260,360 (crop names).  Host code: n is tens to hundreds of boxes per slide.
Pinned by golden vectors produced by running the reference's own class (tests/golden/make_golden_merge.py).

A rectangle is a list [x1, y1, x2, y2, confidence, area, overlap_with_candidate].
"""
import csv
import os
import sys
import time
from argparse import ArgumentParser

UNCONDITIONAL_MERGE_THRESHOLD = 0.6      # merge_overlaped_glomus.py:31
SIDE_LENGTH_MERGE_THRESHOLD = 30         # micrometres, :33
MAX_GLOMUS_SIZE = 350.0                  # micrometres, :37
MAX_GLOMUS_AREA = 300.0 * 300.0          # square micrometres, :38
MAGNIFICATION = 8                        # make_seg_data.py: level-0 -> file-name coordinates


def overlap_area(r1, r2):
    """Intersection area; touching edges count as overlapping with area 0 (:292-301)."""
    if r1[2] >= r2[0] and r1[0] <= r2[2] and r1[3] >= r2[1] and r1[1] <= r2[3]:
        return (min(r1[2], r2[2]) - max(r1[0], r2[0])) * (min(r1[3], r2[3]) - max(r1[1], r2[1]))
    return 0.0


def should_merge(r1, r2, area1, area2, ov, mpp_x, mpp_y, overlap_threshold):
    """merge_decision (:304-339), rule order preserved."""
    if ov >= area1 * UNCONDITIONAL_MERGE_THRESHOLD and ov >= area2 * UNCONDITIONAL_MERGE_THRESHOLD:
        return True
    near = SIDE_LENGTH_MERGE_THRESHOLD
    dx1, dx2 = abs(r1[0] - r2[0]) * mpp_x, abs(r1[2] - r2[2]) * mpp_x
    dy1, dy2 = abs(r1[1] - r2[1]) * mpp_y, abs(r1[3] - r2[3]) * mpp_y
    if dx1 < near and dx2 < near and (dy1 < near or dy2 < near):
        return True
    if dy1 < near and dy2 < near and (dx1 < near or dx2 < near):
        return True
    if max(r1[2] - r1[0], r2[2] - r2[0]) > MAX_GLOMUS_SIZE / mpp_x or max(r1[3] - r1[1], r2[3] - r2[1]) > MAX_GLOMUS_SIZE / mpp_y:
        return False
    if max(area1, area2) > MAX_GLOMUS_AREA / mpp_x / mpp_y:
        return False
    return max(ov / area1, ov / area2) >= overlap_threshold


def try_merge(rect, new_rect, mpp_x, mpp_y, overlap_threshold):
    """merge_rect (:263-289): bounding box of the pair with the larger confidence, or None."""
    ov = overlap_area(new_rect, rect)
    if not ov > 0.0:
        return None
    a1 = (rect[2] - rect[0]) * (rect[3] - rect[1])
    a2 = (new_rect[2] - new_rect[0]) * (new_rect[3] - new_rect[1])
    if not should_merge(rect, new_rect, a1, a2, ov, mpp_x, mpp_y, overlap_threshold):
        return None
    x1, y1 = min(new_rect[0], rect[0]), min(new_rect[1], rect[1])
    x2, y2 = max(new_rect[2], rect[2]), max(new_rect[3], rect[3])
    return [x1, y1, x2, y2, max(new_rect[4], rect[4]), (x2 - x1) * (y2 - y1), 0.0]


def _remerge(kept, merged, mpp_x, mpp_y, overlap_threshold):
    """recheck_overlap (:240-261).  Quirk kept: every kept rectangle that merges with `merged` is
    dropped, but the value handed back is the result for the LAST kept rectangle only (None when
    that one did not merge, even if earlier ones did and were dropped)."""
    result = None
    drop = []
    for i, rect in enumerate(kept):
        result = try_merge(rect, merged, mpp_x, mpp_y, overlap_threshold)
        if result is not None:
            drop.append(i)
    for i in reversed(drop):
        kept.pop(i)
    return result


def insert(rect_list, new_rect, mpp_x, mpp_y, overlap_threshold):
    """check_overlap (:185-228): returns the new list of rectangles after inserting one detection."""
    for rect in rect_list:
        rect[6] = overlap_area(new_rect, rect)
    ordered = sorted(rect_list, key=lambda r: float(r[6]), reverse=True)   # stable, like the reference
    out = []
    merged_any = False
    for rect in ordered:
        merged = try_merge(rect, new_rect, mpp_x, mpp_y, overlap_threshold)
        if merged is None:
            out.append(rect)
            continue
        again = _remerge(out, merged, mpp_x, mpp_y, overlap_threshold)
        if again is not None:
            merged = again
        out.append(merged)
        merged_any = True
        new_rect = merged
    if not merged_any:
        out.append(new_rect)
    return out


def merge_detections(dets, mpp_x, mpp_y, overlap_threshold, conf_threshold=0.6):
    """check_overlap_from_list (:168-183) over one slide's detections.
    dets: iterable of (x1, y1, x2, y2, confidence); returns merged [x1, y1, x2, y2, conf, area, 0.0] lists."""
    cand = []
    for d in dets:
        x1, y1, x2, y2, conf = (float(v) for v in d[:5])
        if conf >= conf_threshold:                       # :144
            cand.append([x1, y1, x2, y2, conf, (x2 - x1) * (y2 - y1), 0.0])
    cand.sort(key=lambda r: float(r[5]), reverse=True)   # largest area first, stable (:180)
    rects = []
    for r in cand:
        rects = insert(rects, r, mpp_x, mpp_y, overlap_threshold)
    return rects


# --------------------------------------------------------------------------- file contracts
def read_detections_csv(path):
    """Detections CSV (detect_glomus_test.py:319-325): site, specimen, file, 'new', ISO time,
    x1, y1, x2, y2, score.  Yields consecutive per-file groups like the reference's reader loop
    (merge_overlaped_glomus.py:100-150): (site, specimen, file, [[x1,y1,x2,y2,score], ...])."""
    groups = []
    with open(path, "r") as f:
        for row in csv.reader(f):
            if not groups or groups[-1][2] != row[2]:
                groups.append((row[0], row[1], row[2], []))
            groups[-1][3].append([float(v) for v in row[5:10]])
    return groups


def merged_csv_rows(site, specimen, file_name, rects):
    """Merged-list rows (merge_overlaped_glomus.py:121-124): site,specimen,"file",x1,y1,x2,y2,conf."""
    return [site + ',' + specimen + ',"' + file_name + '",' + str(int(r[0])) + ',' + str(int(r[1])) + ',' +
            str(int(r[2])) + ',' + str(int(r[3])) + ',' + str(r[4]) + '\n' for r in rects]


def read_merged_csv(path):
    """make_seg_data.py:248-260: {specimen without spaces: [[x1,y1,x2,y2,conf], ...]} in file order."""
    out, order = {}, []
    with open(path, "r") as f:
        for row in csv.reader(f):
            key = row[1].replace(' ', '')
            if key not in out:
                out[key] = []
                order.append(key)
            out[key].append([int(row[3]), int(row[4]), int(row[5]), int(row[6]), float(row[7])])
    return out, order


def crop_name(rect):
    """make_seg_data.py:360: name of the level-0 crop of a merged box (coordinates / 8)."""
    return "xmin{}_ymin{}_xmax{}_ymax{}".format(int(rect[0] / MAGNIFICATION), int(rect[1] / MAGNIFICATION),
                                                 int(rect[2] / MAGNIFICATION), int(rect[3] / MAGNIFICATION))


def merge_csv(detected_csv, merged_csv, mpp_of, overlap_threshold, conf_threshold=0.6):
    """The run() loop: detections CSV -> merged CSV.  mpp_of(specimen, file) -> (mpp_x, mpp_y)."""
    with open(merged_csv, "w") as out:
        for site, specimen, file_name, dets in read_detections_csv(detected_csv):
            mpp_x, mpp_y = mpp_of(specimen, file_name)
            rects = merge_detections(dets, mpp_x, mpp_y, overlap_threshold, conf_threshold)
            out.writelines(merged_csv_rows(site, specimen, file_name, rects))


# --------------------------------------------------------------------------- command line
PNG_EXT = ('.png', '.PNG')


def read_target_list(path):
    """run() (:62-95): {file id (second path element of a line's first field): metadata}; short lines zero the metadata."""
    out = {}
    if path and os.path.isfile(path):
        with open(path, 'r') as f:
            for line in f.readlines():
                parts = line.strip().split(',')
                meta = dict(org_slide_width=0, org_slide_height=0, org_slide_objective_power=0.0, slide_downsample=0.0, mpp_x=0.0, mpp_y=0.0)
                if len(parts) >= 7:
                    meta = dict(org_slide_width=int(parts[1]), org_slide_height=int(parts[2]), org_slide_objective_power=float(parts[3]),
                                slide_downsample=float(parts[4]), mpp_x=float(parts[5]), mpp_y=float(parts[6]))
                ids = parts[0].split('/')
                if len(ids) > 1:
                    out[ids[1]] = meta
    return out


def mpp_lookup(target_list, annotation_dir, staining_dir):
    """check_mpp (:342-362): PNG slides take mpp from the target list (keyed by the file name without extension); any other
    slide is opened with OpenSlide.  OpenSlide is not part of this stack: without it a non-PNG slide falls back on the target
    list when the list names it, and fails loudly otherwise."""
    def mpp_of(specimen, file_name):
        body, ext = os.path.splitext(file_name)
        if ext not in PNG_EXT:
            try:
                import openslide
            except ImportError:
                openslide = None
            if openslide is not None:
                with openslide.open_slide(os.path.join(annotation_dir, staining_dir, specimen, file_name)) as slide:
                    return (float(slide.properties[openslide.PROPERTY_NAME_MPP_X]), float(slide.properties[openslide.PROPERTY_NAME_MPP_Y]))
            if body not in target_list or not target_list[body]['mpp_x'] > 0:
                raise RuntimeError("%s: reading the resolution of a non-PNG slide needs the openslide package (not installed) or a "
                                   "target-list line with its metadata" % file_name)
        props = target_list[body]          # KeyError for an unknown file, as in the reference
        return float(props['mpp_x']), float(props['mpp_y'])
    return mpp_of


def build_parser():
    """argparse surface of merge_overlaped_glomus.py:364-383."""
    p = ArgumentParser(description='MERGE_OVERLAPPED_GLOMUS')
    p.add_argument('--staining', dest='staining', type=str, default='OPT_PAS')
    p.add_argument('--target_list', dest='target_list', type=str)
    p.add_argument('--detected_list', dest='input_file', type=str)
    p.add_argument('--output_dir', dest='output_dir', type=str)
    p.add_argument('--output_file_ext', dest='training_type', type=str, default='')
    p.add_argument('--conf_threshold', dest='conf_threshold', type=float, default=0.6)
    p.add_argument('--data_dir', dest='annotation_dir', type=str)
    p.add_argument('--overlap_threshold', dest='overlap_threshold', type=float)
    return p


def run(args, out=sys.stdout):
    """MargeOverlapedGlomus.run (:56-166): <staining>_GlomusMergedList_<ext>.csv and ..._log.csv in --output_dir."""
    from .detect import staining_dir
    target_list = read_target_list(args.target_list)
    mpp_of = mpp_lookup(target_list, args.annotation_dir or '', staining_dir(args.staining))
    body = args.staining + '_GlomusMergedList_' + args.training_type
    merged_path = os.path.join(args.output_dir, body + '.csv')
    log_path = os.path.join(args.output_dir, body + '_log.csv')
    with open(merged_path, "w") as merged, open(log_path, "w") as log:
        t0 = time.time()
        for site, specimen, file_name, dets in read_detections_csv(args.input_file):
            mpp_x, mpp_y = mpp_of(specimen, file_name)
            rects = merge_detections(dets, mpp_x, mpp_y, args.overlap_threshold, args.conf_threshold)
            merged.writelines(merged_csv_rows(site, specimen, file_name, rects))
            merged.flush()
            print('"{}":{}'.format(file_name, rects), file=out)
            log.write('"{}",{}\n'.format(file_name, time.time() - t0))
            log.flush()
            t0 = time.time()
    return merged_path


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.input_file or not args.output_dir or args.overlap_threshold is None:
        print("--detected_list, --output_dir and --overlap_threshold are required", file=sys.stderr)
        return 2
    run(args)
    return 0


if __name__ == '__main__':
    sys.exit(main())
