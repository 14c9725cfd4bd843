"""Test helper (one rank of a sharded segment command line without a GPU): segment.py's own argument parser, list sharding,
evaluate() and file writing, with a CPU stand-in behind the engine (an ESPNet-C-shaped stub: its maps come from the
segment_images stand-in below, counts and overlays from the host arithmetic) -- what main() does after it has made the engine.
The rank named by GS_TEST_FAIL_RANK raises inside its loop, alone."""
import glob
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glomeruli_segmentation_amd import segment  # noqa: E402
from glomeruli_segmentation_amd.shard import abort_rank, finish_ranks, init_from_env, rank_range  # noqa: E402


def fake_segment_images(engine, images, mean, std, width, height, batch, want_net_maps=False):
    if os.environ.get("RANK", "0") == os.environ.get("GS_TEST_FAIL_RANK"):
        raise RuntimeError("this rank fails on purpose")
    outs, nets = [], []
    for im in images:
        cm = (im[:, :, 0].astype(np.int32) // 52).astype(np.uint8) % 5
        ys = (np.arange(height) * im.shape[0] // height)
        xs = (np.arange(width) * im.shape[1] // width)
        outs.append(cm)
        nets.append(np.ascontiguousarray(cm[ys][:, xs]))
    return (outs, nets) if want_net_maps else outs


def main(argv):
    args = segment.build_parser().parse_args(argv)
    if args.overlay:
        args.colored = True
    rgb_list = sorted(glob.glob(args.rgb_data_dir + "/*/*.PNG"))
    rank, world, local, dist = init_from_env(use_gpu=False)
    if args.workers is None:
        args.workers = 1
    lo, hi = rank_range(len(rgb_list), rank, world)
    print("rank %d of %d: crops [%d, %d)" % (rank, world, lo, hi))
    segment.segment_images = fake_segment_images
    engine = types.SimpleNamespace(encoder_only=True, classes=5, device=None)
    try:
        segment.evaluate(args, engine, rgb_list[lo:hi], [None] * (hi - lo), rank, world, dist)
    except BaseException:
        abort_rank(dist)
        raise
    finish_ranks(dist)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
