#!/usr/bin/env python3
"""The segment command line end to end (module/espnet/test/VisualizeResults_iou.py:100-182 as
`python -m glomeruli_segmentation_amd.segment`): PNG crops on disk in -> overlay JPEG, original PNG, class map PNG, labelme
JSON (with the base64 crop) and summary_pixel.csv out, with the reference README's flags (--colored --overlay --cityFormat).

    python tools/bench_cli.py [--crops 448] [--out profiles/r04_cli_bench.json]

448 crops of the example slide's 28 box sizes (mean 0.53 Mpx), written once as PNG under a scratch directory; the command is
run with --workers 0 (the serial loop: decode, GPU, encode / trace / write one after the other, the reference's structure),
with the default worker pool (decode-ahead / write-behind while a batch is on the GPU) and with TWO workers on two CPUs -- a
rank's share of the cores on an 8-GPU node.  The work is host-bound -- PNG decode and three PNG / JPEG encodes per crop against
~0.1 ms of GPU time -- so the figure scales with the cores the process has; the core count is in the line.  The serial run also
gives the per-stage breakdown of the host time (segment.STAGE_SECONDS): decode / GPU pass (counts and overlays included) /
overlay JPEG / original PNG / class-map PNG / contours / base64 PNG / JSON."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=448)
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    from PIL import Image
    from glomeruli_segmentation_amd import segment
    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile
    root = tempfile.mkdtemp(prefix="glomseg_cli_")
    try:
        ex = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
        base = [synth_tile(7000 + k, int(b[3] - b[1]), int(b[2] - b[0]), blobs=4) for k, b in enumerate(ex)]
        rgb = os.path.join(root, "org_image")
        t0 = time.perf_counter()
        for i in range(a.crops):
            d = os.path.join(rgb, "slide%02d" % (i // 56))
            os.makedirs(d, exist_ok=True)
            Image.fromarray(np.ascontiguousarray(base[i % len(base)][:, :, ::-1])).save(
                os.path.join(d, "xmin%d_ymin%d_xmax%d_ymax%d.PNG" % (i, i, i + 100, i + 100)), compress_level=1)
        t_gen = time.perf_counter() - t0
        z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
        wpath = os.path.join(root, "fold1.npz")
        np.savez(wpath, **{k: z[k] for k in z.files})
        mean, std = FOLD_MEAN_STD[1]
        common = ["--rgb_data_dir", rgb, "--weights", wpath, "--gpu_id", "0", "--classes", "5", "--img_extn", "PNG", "--decoder",
                  "--colored", "--overlay", "--cityFormat", "--mean"] + [str(v) for v in mean] + ["--std"] + [str(v) for v in std]
        res = {}
        cores = segment.default_workers()
        allowed = sorted(os.sched_getaffinity(0))
        stages = None
        for tag, workers, cpus in (("warmup", cores, None), ("serial", 0, None), ("overlapped", cores, None), ("two_workers_two_cpus", 2, allowed[:2])):
            out = os.path.join(root, "out_" + tag)
            if cpus is not None:
                os.sched_setaffinity(0, cpus)
            segment.STAGE_SECONDS = {} if tag == "serial" else None
            t0 = time.perf_counter()
            rc = segment.main(common + ["--savedir", out, "--workers", str(workers)])
            el = time.perf_counter() - t0
            if cpus is not None:
                os.sched_setaffinity(0, allowed)
            assert rc == 0
            n_files = sum(len(fs) for _, _, fs in os.walk(out))
            res[tag] = {"workers": workers, "cpus": len(cpus) if cpus is not None else len(allowed), "seconds": round(el, 3),
                        "cli_crops_per_s": round(a.crops / el, 1), "files_written": n_files}
            if tag == "serial":
                stages = dict(segment.STAGE_SECONDS)
                segment.STAGE_SECONDS = None
                acc = sum(stages.values())
                res[tag]["stage_ms_per_crop"] = {k: round(1e3 * v / a.crops, 2) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])}
                res[tag]["stage_ms_per_crop"]["(unaccounted: engine creation, weight load, CSV, thread hand-off)"] = round(1e3 * (el - acc) / a.crops, 2)
            print(tag, res[tag], flush=True)
        same = True
        import filecmp
        for d, _, fs in os.walk(os.path.join(root, "out_serial")):
            for f in fs:
                p1 = os.path.join(d, f)
                for other in ("out_overlapped", "out_two_workers_two_cpus"):
                    p2 = p1.replace("out_serial", other)
                    same = same and os.path.isfile(p2) and filecmp.cmp(p1, p2, shallow=False)
        line = {"what": "python -m glomeruli_segmentation_amd.segment end to end (engine creation included), %d PNG crops of the example "
                        "slide's sizes -> overlay jpg + org png + class map png + labelme json (base64 crop) + summary_pixel.csv; "
                        "--colored --overlay --cityFormat --decoder" % a.crops,
                "crops": a.crops, "host_cores_of_the_process": cores, "mean_crop_px": int(np.mean([c.shape[0] * c.shape[1] for c in base])),
                "serial": res["serial"], "overlapped": res["overlapped"], "two_workers_two_cpus": res["two_workers_two_cpus"],
                "speedup": round(res["serial"]["seconds"] / res["overlapped"]["seconds"], 2),
                "outputs_identical_byte_for_byte": bool(same), "png_generation_s": round(t_gen, 2)}
        print(json.dumps(line))
        if a.out:
            with open(a.out, "w") as f:
                json.dump(line, f, indent=1)
    finally:
        if not a.keep:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
