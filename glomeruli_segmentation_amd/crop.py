#!/usr/bin/env python3
"""Level-0 crops of the merged detections: the command line of module/faster-rcnn/make_seg_data.py (argparse :364-382,
`read_detected_glomus_list` :248-260, `output_org_files` :347-361) -- the step between the merge and the segmentation.

    python -m glomeruli_segmentation_amd.crop --staining OPT_PAS --target_list T.txt --merged_detection_result_csv M.csv \
        --wsi_dir DATA/02_PAS --output_dir OUT/seg_data

For every merged box of every slide: the level-0 region (x1, y1, x2 - x1, y2 - y1) as
`<output_dir>/org_image/<slide>/xmin{x1/8}_ymin{y1/8}_xmax{x2/8}_ymax{y2/8}.PNG` (:357-361), which is what
`python -m glomeruli_segmentation_amd.segment --rgb_data_dir <output_dir>/org_image` globs.  Host I/O only; nothing here runs
on the GPU (the resampling of the crops to the network size does: gs_espnet_segment_crops_host).

Slides: `.ndpi` through OpenSlide when that is installed (`read_region` at level 0, RGBA, saved as the reference saves it).
Without OpenSlide -- this image -- the PNG branch of the detector's command line is followed (detect_glomus_test.py:170-173,
196-234): the slide is a PNG at 1/downsample of level 0 under <wsi_dir>/<slide>/, its downsample stands in the target list's
metadata line, and a level-0 region is that PNG sampled at floor(level-0 coordinate / downsample) -- a stand-in for the
pyramid's level 0 that keeps every file contract (names, sizes, channel order).  The ground-truth branch (`scan_files`:
annotation XML + labelme shapes -> label PNGs, :388-392) needs the GT tooling that is out of scope:
its three directories are parsed, and refused with an explanation when the reference would take that branch.
"""
import glob
import os
import sys
from argparse import ArgumentParser

import numpy as np

from . import detect, merge


def build_parser():
    """argparse surface of make_seg_data.py:364-382, every flag with the reference's dest and default"""
    p = ArgumentParser(description='Make segmentation data from the result of the detection')
    p.add_argument('--staining', dest='staining', type=str, required=True)
    p.add_argument('--merged_detection_result_csv', dest='input_csv', type=str, required=True)
    p.add_argument('--target_list', dest='target_list', type=str, required=True)
    p.add_argument('--wsi_dir', dest='wsi_dir', type=str, required=True)
    p.add_argument('--segmentation_gt_json_dir', dest='seg_gt_json_dir', type=str, default=None)
    p.add_argument('--object_detection_gt_xml_dir', dest='ob_gt_xml_dir', type=str, default=None)
    p.add_argument('--iou_threshold', dest='iou_threshold', type=float, default=0.01)
    p.add_argument('--output_dir', dest='output_dir', type=str, default='./output/seg_data')
    p.add_argument('--start', dest='start', type=int, default=0)
    p.add_argument('--end', dest='end', type=int, default=0)
    p.add_argument('--segmentation_gt_png_dir', dest='gt_png_dir', type=str, default=None)
    p.add_argument('--no_save', dest='no_save', action='store_true')
    return p


class PngSlide:
    """a slide given as a PNG at 1/downsample of level 0 (the detector's PNG branch): read_region at level 0 = the PNG
    sampled at floor(level-0 coordinate / downsample), clamped to the image (OpenSlide pads with transparent black
    beyond the slide; a merged box never leaves it)"""

    def __init__(self, path, downsample):
        from PIL import Image
        with Image.open(path) as im:
            self.rgb = np.asarray(im.convert("RGB"))
        self.ds = float(downsample)

    def read_region(self, x, y, w, h):
        ys = np.clip(((y + np.arange(h)) / self.ds).astype(np.int64), 0, self.rgb.shape[0] - 1)
        xs = np.clip(((x + np.arange(w)) / self.ds).astype(np.int64), 0, self.rgb.shape[1] - 1)
        rgba = np.empty((h, w, 4), dtype=np.uint8)
        rgba[:, :, :3] = self.rgb[ys][:, xs]
        rgba[:, :, 3] = 255
        return rgba


def open_slide(wsi_dir, file_key, target_meta):
    """(read_region(x, y, w, h) -> uint8 RGBA [h,w,4], description) for the slide of `file_key` (:349-352)"""
    ndpi = glob.glob(os.path.join(wsi_dir, file_key, "*ndpi"))
    if ndpi:
        try:
            import openslide
        except ImportError:
            openslide = None
        if openslide is not None:
            assert len(ndpi) == 1                                              # :350
            slide = openslide.open_slide(ndpi[0])
            return (lambda x, y, w, h: np.asarray(slide.read_region((x, y), 0, (w, h)))), ndpi[0]
    pngs = sorted(glob.glob(os.path.join(wsi_dir, file_key, "*.PNG")) + glob.glob(os.path.join(wsi_dir, file_key, "*.png")))
    meta = target_meta.get(file_key)
    if pngs and meta and meta["downsample"] > 0:
        s = PngSlide(pngs[0], meta["downsample"])
        return s.read_region, pngs[0]
    raise RuntimeError("slide %s: no .ndpi readable (OpenSlide is not installed) and no PNG slide with a metadata line "
                       "(id/file,w,h,power,downsample,mppx,mppy) in the target list" % file_key)


def output_org_files(args, out=sys.stdout):
    """output_org_files (:347-361): one PNG per merged box, named by its level-0 coordinates / 8"""
    from PIL import Image
    boxes_of, order = merge.read_merged_csv(args.input_csv)                    # :248-260
    target_meta = {}
    if os.path.isfile(args.target_list):
        for line in open(args.target_list):
            m = detect.parse_target_line(line)
            if m:
                target_meta[m["specimen_id"].replace(' ', '')] = m
    if not os.path.isdir(args.output_dir):
        os.makedirs(args.output_dir)
    written = {}
    for key in order:
        read_region, what = open_slide(args.wsi_dir, key, target_meta)
        odir = os.path.join(args.output_dir, "org_image", key)
        if not os.path.exists(odir):
            os.makedirs(odir)
        names = []
        for b in boxes_of[key]:
            region = read_region(b[0], b[1], b[2] - b[0], b[3] - b[1])         # :358
            name = merge.crop_name(b)                                          # :360
            # (written whatever --no_save says: the reference reads that flag in its ground-truth branch only, scan_files;
            # output_org_files, make_seg_data.py:347-361, always saves)
            Image.fromarray(region).save(os.path.join(odir, name + '.PNG'), format="PNG")     # :361
            names.append(name)
        written[key] = names
        print("{}: {} crops from {}".format(key, len(names), what), file=out)
    return written


def main(argv=None):
    args = build_parser().parse_args(argv)
    # the reference takes the ground-truth branch unless the JSON or the XML directory is missing (:388)
    if args.seg_gt_json_dir is not None and args.ob_gt_xml_dir is not None:
        print("the ground-truth branch (scan_files: annotation XML + labelme shapes -> label PNGs) is outside the rebuilt path; "
              "leave --segmentation_gt_json_dir / --object_detection_gt_xml_dir unset to write the crops only", file=sys.stderr)
        return 2
    output_org_files(args)
    return 0


if __name__ == '__main__':
    sys.exit(main())
