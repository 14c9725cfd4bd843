"""CPU tests of the host side: C-ABI exports, drop-in Model shim, sharding (gloo, 2 ranks),
driver helpers.  No GPU compute is called."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden, load_weights


def test_cabi_exports_every_declared_symbol():
    from glomeruli_segmentation_amd import _lib
    from glomeruli_segmentation_amd.build import build_lib
    build_lib()
    header = open(os.path.join(REPO, "include", "glomseg.h")).read()
    declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = _lib.load()                       # resolves every prototype or raises
    assert lib.gs_abi_version() == _lib.ABI_VERSION
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (gs_[a-z0-9_]+)", out))
    assert declared <= exported


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from glomeruli_segmentation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_pack_state_dict_layout(sd1):
    from glomeruli_segmentation_amd.engine import pack_state_dict
    blob, table = pack_state_dict(sd1)
    assert len(table) == sum(1 for v in sd1.values() if v.dtype.kind == "f")     # int64 counters skipped
    names = {t.name.decode(): t for t in table}
    t = names["encoder.level3.7.d16.conv.weight"]
    assert list(t.shape) == [25, 25, 3, 3] and t.ndim == 4
    assert np.array_equal(blob[t.offset:t.offset + 25 * 25 * 9], sd1["encoder.level3.7.d16.conv.weight"].ravel())
    assert blob.size == sum(v.size for v in sd1.values() if v.dtype.kind == "f")


def test_model_shim_state_dict_is_drop_in(sd1):
    """same parameter tree as the reference: every key of espnet_fold1.pth loads, none is left over"""
    import torch
    import glomeruli_segmentation_amd.Model as Net
    net = Net.ESPNet(5, 2, 8)
    msg = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd1.items()})
    assert not msg.missing_keys and not msg.unexpected_keys
    assert set(net.state_dict().keys()) == set(sd1.keys()) and len(sd1) == 205
    assert isinstance(net.modules, list) and len(net.modules) == 11          # the reference's shadowing list
    enc = Net.ESPNet_Encoder(5, 2, 8)
    enc.load_state_dict({k[8:]: torch.from_numpy(v) for k, v in sd1.items() if k.startswith("encoder.")})
    assert Net.ESPNet().encoder.level3.__len__() == 3 and Net.ESPNet_Encoder().level2.__len__() == 5   # ctor defaults


def test_model_shim_cpu_is_refused_unless_opted_in(sd1, monkeypatch):
    import torch
    import glomeruli_segmentation_amd.Model as Net
    net = Net.ESPNet(5, 2, 8)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd1.items()})
    net.eval()
    z = load_golden("stages_fold1.npz")
    x = torch.from_numpy(z["input"][None])
    monkeypatch.delenv("GLOMSEG_ALLOW_TORCH_CPU", raising=False)
    with pytest.raises(RuntimeError):
        net(x)
    monkeypatch.setenv("GLOMSEG_ALLOW_TORCH_CPU", "1")
    with torch.no_grad():
        out = net(x)[0].numpy()
    assert np.abs(out - z["logits"]).max() <= 1e-4      # the shim's torch graph is the reference's graph


def test_rank_range_partitions_exactly():
    from glomeruli_segmentation_amd.shard import plan_grid, rank_range
    for total in (0, 1, 7, 36, 1369):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = rank_range(total, r, world)
                cover.extend(range(lo, hi))
            assert cover == list(range(total))
    grid = plan_grid(1000, 2300, 512, 1024)
    assert grid[0] == (0, 0) and grid[-1] == (488, 1276) and len(grid) == 2 * 3
    with pytest.raises(ValueError):
        rank_range(4, 2, 2)


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["GS_REPO"])
from glomeruli_segmentation_amd.shard import segment_sharded
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile
from oracle import espnet_oracle as orc

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
z = np.load(os.path.join(os.environ["GS_REPO"], "tests", "golden", "weights_fold1.npz"))
sd = {k: z[k] for k in z.files}
mean, std = FOLD_MEAN_STD[1]
TOTAL = int(os.environ.get("GS_TOTAL", "5"))

def load(lo, hi):
    return np.stack([synth_tile(100 + i, 32, 64, blobs=2) for i in range(lo, hi)])

def compute(tiles):          # the checker stands in for the GPU pass in this CPU test
    res = [orc.segment_tile(t, sd, mean, std) for t in tiles]
    return np.stack([r[1] for r in res]), np.stack([r[2] for r in res])

masks, counts = segment_sharded(compute, load, TOTAL, rank, world, dist=dist, batch=2)
if rank == 0:
    np.savez(os.environ["GS_OUT"], masks=masks, counts=counts)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_sharding_equals_single_process(tmp_path):
    """world_size-2 gloo run gives byte-identical masks and counts to one process"""
    from glomeruli_segmentation_amd.shard import segment_sharded
    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD, synth_tile
    from oracle import espnet_oracle as orc
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    out = tmp_path / "out.npz"
    env = dict(os.environ, GS_REPO=REPO, GS_OUT=str(out), MASTER_ADDR="127.0.0.1", MASTER_PORT="29617",
               WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    sd = load_weights(1)
    mean, std = FOLD_MEAN_STD[1]

    def compute(tiles):
        res = [orc.segment_tile(t, sd, mean, std) for t in tiles]
        return np.stack([r[1] for r in res]), np.stack([r[2] for r in res])

    masks, counts = segment_sharded(compute, lambda lo, hi: np.stack([synth_tile(100 + i, 32, 64, blobs=2) for i in range(lo, hi)]),
                                    5, 0, 1)
    assert np.array_equal(got["masks"], masks) and np.array_equal(got["counts"], counts)
    assert counts.sum() == 5 * 32 * 64
    # more ranks than tiles: rank 0's range is empty (rank_range(1, 0, 2) == (0, 0)); the exchange still completes
    out1 = tmp_path / "out1.npz"
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), GS_TOTAL="1", GS_OUT=str(out1), MASTER_PORT="29619"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got1 = np.load(out1)
    assert np.array_equal(got1["masks"], masks[:1]) and np.array_equal(got1["counts"], np.bincount(masks[0].ravel(), minlength=5))


def test_bench_spawns_one_process_per_gpu():
    """`bench.py --gpus 2` started plainly (no torchrun) spawns two ranks before any GPU call and relays ONE JSON line
    with n_gpus 2, per-rank times and the all-reduced totals (--dry-run: control flow only, gloo, no device work)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["config"]["global_batch"] == 64
    assert len(j["per_rank_ms_per_step"]) == 2 and j["repeats"]["n"] == 5
    assert j["pixel_totals_all_ranks"] == [2 * 3] * 5          # each dry step counts one pixel per class per rank
    assert len(j["host_pipeline"]["per_rank_patches_per_s"]) == 2
    # a world size that contradicts --gpus is refused
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_product_library_has_no_diagnostic_switches():
    """the shipped .so reads no environment: the timing / stamp variants and their knobs exist in -DGS_DIAG builds only"""
    from glomeruli_segmentation_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"GS_VARIANT", b"GS_PRIO", b"GS_STAGGER", b"GS_NO_VEC", b"gpurun_out", b"stamps"):
        assert name not in blob, name
    # and in the sources every getenv sits inside an `#ifdef GS_DIAG` region
    csrc = os.path.join(REPO, "glomeruli_segmentation_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h", ".cpp")):
            continue
        stack = []
        for line in open(os.path.join(csrc, fn)):
            t = line.strip()
            if t.startswith(("#ifdef", "#ifndef", "#if ")):
                stack.append(t)
            elif t.startswith("#endif"):
                stack.pop()
            elif "getenv" in t and not t.startswith("//"):
                assert any("GS_DIAG" in c and c.startswith(("#ifdef", "#if ")) for c in stack), (fn, t)


def test_weight_fixture_digest(sd1):
    """the SHA-256 over (key, fp32 bytes) of the reference state_dict, written when the goldens were made,
    pins the weight fixture that every parity test (and pack_state_dict) starts from"""
    import hashlib
    for fold in range(1, 6):
        sd = load_weights(fold)
        h = hashlib.sha256()
        for k, v in sd.items():
            if v.dtype == np.float32:
                h.update(k.encode())
                h.update(np.ascontiguousarray(v).tobytes())
        want = bytes(load_golden("masks_fold%d.npz" % fold)["weights_sha256"].tolist()).hex()
        assert h.hexdigest() == want, fold


def test_metrics_relabel_and_palette_match_reference_golden():
    """segment.confusion / metric_right against the reference's iouEval (addBatch, getMetricRight), imageops.relabel_city against
    the reference's relabel cascade on all 256 byte values, and the colour table: tests/golden/misc.npz (make_golden_misc.py)"""
    from glomeruli_segmentation_amd import imageops, segment
    z = load_golden("misc.npz")
    total = np.zeros((5, 5), dtype=np.int64)
    for k in range(3):
        h = segment.confusion(z["pred_%d" % k].ravel(), z["gt_%d" % k].ravel(), 5)
        assert np.array_equal(h, z["hist_%d" % k]), k
        total += h
    assert np.array_equal(total, z["total_hist"])
    o, pa, pi, m = segment.metric_right(total)
    assert o == float(z["overall_acc"]) and m == float(z["miou"])
    assert np.array_equal(pa, z["per_class_acc"]) and np.array_equal(pi, z["per_class_iu"])
    assert np.array_equal(imageops.relabel_city(z["relabel_in"]), z["relabel_out"])
    assert np.array_equal(imageops.PALETTE, z["palette"])


def test_image_helpers():
    from glomeruli_segmentation_amd import imageops
    rng = np.random.default_rng(0)
    city = imageops.relabel_city(np.arange(5, dtype=np.uint8))
    assert city.tolist() == [7, 8, 11, 12, 13]                                       # VisualizeResults_iou.py:54-81
    a = np.full((2, 2, 3), 100, dtype=np.uint8)
    b = np.full((2, 2, 3), 201, dtype=np.uint8)
    assert imageops.add_weighted(a, 0.4, b, 0.6)[0, 0, 0] == 161                   # round(40 + 120.6)
    cm = rng.integers(0, 5, (6, 10)).astype(np.uint8)
    col = imageops.colourise(cm)
    assert col.shape == (6, 10, 3) and (col[cm == 1] == [0, 0, 255]).all()           # class 1 is red, stored BGR


def test_image_oracle_matches_independent_resize_fixture():
    """the oracle's restatement of cv2.resize (INTER_LINEAR on float32, INTER_NEAREST) against outputs of
    torch.nn.functional.interpolate (tests/golden/make_golden_resize.py)"""
    from oracle import image_oracle as io
    z = load_golden("resize.npz")
    mean, std = z["mean"], z["std"]
    for k, (h, w, oh, ow) in enumerate(z["cases"].tolist()):
        got = io.normalise_then_resize(z["crop_%d" % k], mean, std, ow, oh)
        assert got.shape == z["net_%d" % k].shape
        assert np.abs(got - z["net_%d" % k]).max() <= 2e-6, k
        assert np.array_equal(io.resize_nearest(z["cmap_%d" % k], w, h), z["back_%d" % k]), k
    img = np.random.default_rng(0).random((6, 10, 3)).astype(np.float32)
    assert np.array_equal(io.resize_linear_f32(img, 10, 6), img)                     # identity at equal size


def test_reference_window_luts():
    """the compositor's sample tables against the oracle's window walk on a 1-D ramp (no GPU needed)"""
    from glomeruli_segmentation_amd.composite import reference_window_luts
    from oracle import image_oracle as io
    for W, H in [(5003, 3100), (3100, 5003), (7200, 4800), (2399, 2399)]:
        sx, sy = reference_window_luts(W, H)
        assert sx.shape == (int(W / 8),) and sy.shape == (int(H / 8),)
        # a "crop" covering the whole slide whose value encodes the column (mod 251) + 1
        col = (np.arange(W) % 251 + 1).astype(np.uint8)
        full = np.broadcast_to(col, (H, W))
        ref = io.reference_wsi_pred_map([full], [(0, 0, W, H)], W, H)
        exp = np.zeros_like(ref)
        ys, xs = np.nonzero(sy >= 0)[0], np.nonzero(sx >= 0)[0]
        exp[np.ix_(ys, xs)] = col[sx[xs]][None, :]
        assert np.array_equal(exp, ref), (W, H)


def test_driver_flags_match_reference():
    from glomeruli_segmentation_amd.segment import build_parser
    a = build_parser().parse_args(["--rgb_data_dir", "d", "--weights", "w", "--mean", "1", "2", "3", "--std", "1", "1", "1"])
    assert (a.inWidth, a.inHeight, a.modelType, a.p, a.q, a.classes, a.gpu_id, a.img_extn, a.savedir, a.scaleIn) == \
        (1024, 512, 1, 2, 8, 5, -1, "PNG", "./results", 1)
    assert not (a.cityFormat or a.colored or a.overlay or a.decoder)


def test_detector_window_geometry_known_answers():
    """written from detect_glomus_test.py:286-304,255-261 with the example slide of SURVEY 8c:
    53248 x 23040 px (6656 x 2880 at ds 8), mpp 0.2277, --window_size 2000 --overlap_ratio 0.1"""
    import math
    from glomeruli_segmentation_amd import detect
    assert detect.pick_level(40, [1.0, 2.0, 4.0, 8.0, 16.0]) == (3, 8.0)
    assert detect.pick_level(20, [1.0, 2.0, 4.0]) == (2, 4.0)
    assert detect.pick_level(40, [1.0, 2.0]) == (3, 8.0)                    # fallback :255-256
    p = detect.plan_windows(53248, 23040, 0.2277, 0.2277, 8.0, 2000, 0.1)
    assert math.isclose(p.window_x_org, 2000 / 0.2277)
    assert (p.x_split_times, p.y_split_times, p.window_x, p.window_y) == (7, 3, 1098, 1098)
    assert (p.step_x, p.step_y) == (7905, 7905)                              # int(8783.48 * 0.9), level-0 px
    assert p.origins()[8] == (1, 1, 7905, 7905) and len(p.origins()) == 21
    d = detect.plan_windows(40000, 40000, 0.2277, 0.2277, 8.0, None, None)   # defaults 500 um / 0.5
    assert (d.x_split_times, d.window_x, d.step_x) == (37, 275, 1097)
    pi = detect.plan_windows(6656, 2880, 0.2277, 0.2277, 8.0, 2000, 0.1, from_image=True)
    assert pi.step_x == int(1098 * 0.9)                                      # :218 stride in PNG pixels


def test_detector_host_logic_matches_reference_golden():
    """window geometry, strides and CSV rows against tests/golden/detect.npz, written by the reference's own GlomusDetector
    (calc_window_size, the strides of scan_region / scan_region_from_image, write_detected_result with a frozen clock)"""
    import datetime
    from glomeruli_segmentation_amd import detect
    z = load_golden("detect.npz")
    now = datetime.datetime(2020, 1, 2, 3, 4, 5)
    for k, (case, geo, rows) in enumerate(zip(z["cases"], z["geometry"], z["rows"])):
        w, h, mx, my, ds, win, ov = case.tolist()
        win, ov = (None, None) if win < 0 else (int(win), ov)
        p0 = detect.plan_windows(int(w), int(h), mx, my, ds, win, ov)
        pi = detect.plan_windows(int(w), int(h), mx, my, ds, win, ov, from_image=True)
        assert [p0.window_x_org, p0.window_y_org, p0.x_split_times, p0.y_split_times, p0.window_x, p0.window_y, p0.step_x, p0.step_y,
                pi.step_x, pi.step_y] == geo.tolist(), k
        bs = [[10 + k, 20, 300 + 7 * k, 411, np.float32(0.91)], [0, 0, p0.window_x, p0.window_y, np.float32(0.6)], [5, 6, 7, 8, 0.0]]
        got = detect.csv_rows(bs, p0.step_x * 1, p0.step_y * 2, ds, "site_a", "H16-%04d" % k, "H16-%04d_PAS.ndpi" % k, now)
        got += detect.csv_rows(bs[:1], pi.step_x * 3 * ds, pi.step_y * 1 * ds, ds, "site_a", "H16-%04d" % k, "H16-%04d_PAS.PNG" % k, now)
        assert "".join(got) == str(rows), k
    assert [detect.staining_dir(t) for t in ("OPT_PAS", "OPT_PAM", "OPT_MT", "OPT_Azan", "OPT_HE", "x")] == z["types"].tolist()


def test_detector_box_postprocessing_and_csv():
    import datetime
    from glomeruli_segmentation_amd import detect
    boxes = np.array([[[0.1, 0.2, 0.5, 0.6], [0.0, 0.0, 1.0, 1.0], [0.3, 0.3, 0.4, 0.4]]], dtype=np.float32)
    scores = np.array([[0.9, 0.61, 0.2]], dtype=np.float32)
    bs = detect.boxes_from_detector(boxes, scores, 1098, 1000, thresh=0.6)
    assert [b[:4] for b in bs] == [[int(1098 * np.float32(0.2)), 100, int(1098 * np.float32(0.6)), 500], [0, 0, 1098, 1000]]
    # known answer for the promotion rule: the reference's NumPy 1.x forms WINDOW_X * xmin (Python int x float32 scalar) in
    # float64 -- 1098 * float64(float32(4/1098)) = 3.99999998... -> 3 -- where NumPy 2 would stay in float32 and give 4
    edge = np.array([[[np.float32(2 / 1000.0), np.float32(4 / 1098.0), 0.5, 0.5]]], dtype=np.float32)
    assert detect.boxes_from_detector(edge, np.array([[0.7]], dtype=np.float32), 1098, 1000, 0.6)[0][:4] == \
        [int(1098 * float(np.float32(4 / 1098.0))), int(1000 * float(np.float32(2 / 1000.0))), 549, 500] == [3, 2, 549, 500]
    now = datetime.datetime(2020, 1, 2, 3, 4, 5)
    rows = detect.csv_rows(bs, 7905, 0, 8.0, "site", "PAS-001", "PAS-001.ndpi", now)
    x1 = 7905 + bs[0][0] * 8.0
    assert rows[0] == '"site","PAS-001","PAS-001.ndpi",new,2020-01-02T03:04:05,%s,800.0,%s,4000.0,%s\n' % (
        str(x1), str(7905 + bs[0][2] * 8.0), str(bs[0][4]))
    assert detect.boxes_from_detector(boxes, np.zeros((1, 3)), 10, 10, 0.5) == []
    assert detect.parse_target_line("#PAS-001/PAS-001") is None
    m = detect.parse_target_line("PAS-001/PAS-001,53248,23040,40,8,0.2277,0.2277\n")
    assert (m["width"], m["mpp_x"], m["specimen_id"], m["file_name"]) == (53248, 0.2277, "PAS-001", "PAS-001")
    assert detect.parse_target_line("PAS-001/PAS-001")["width"] == 0       # short line zeroes metadata (:115-122)


def test_detector_scan_shards_by_window_range():
    import datetime
    from glomeruli_segmentation_amd import detect
    plan = detect.plan_windows(8000, 4000, 0.25, 0.25, 8.0, 500, 0.5)
    calls = []

    def read_region(x, y, w, h):
        calls.append((x, y))
        return np.zeros((h, w, 4), dtype=np.uint8)

    def detector(im):
        assert im.shape == (1, plan.window_y, plan.window_x, 3) and im.dtype == np.uint8
        return np.array([[[0.0, 0.0, 0.5, 0.5]]]), np.array([[0.7]]), np.array([[1]]), np.array([1])

    now = datetime.datetime(2020, 1, 1)
    whole = detect.scan_slide(read_region, detector, plan, 0.6, "s", "p", "f", now=now)
    parts = []
    for r in range(3):
        parts += detect.scan_slide(read_region, detector, plan, 0.6, "s", "p", "f", rank=r, world=3, now=now)
    assert parts == whole and len(whole) == plan.x_split_times * plan.y_split_times
    a = detect.build_parser().parse_args([])
    assert (a.data_category, a.conf_threshold, a.model_name, a.output_file_ext) == ("OPT_PAM", 0.6, "frozen_inference_graph.pb", "_GlomusList")


def test_merge_matches_reference_golden():
    """greedy merge of overlapping detections vs the reference's own class on 19 golden cases"""
    from glomeruli_segmentation_amd import merge
    z = load_golden("merge.npz")
    total_in = total_out = 0
    for i in range(int(z["n_cases"])):
        mpp, thr = z["par_%d" % i]
        got = merge.merge_detections(z["in_%d" % i], mpp, mpp, thr)
        got = np.array([r[:5] for r in got], dtype=np.float64).reshape(-1, 5)
        assert got.shape == z["out_%d" % i].shape, i
        assert np.array_equal(got, z["out_%d" % i]), i        # same boxes, same order, bit for bit
        total_in += len(z["in_%d" % i])
        total_out += len(got)
    assert total_out < total_in


def test_merge_csv_contracts(tmp_path):
    from glomeruli_segmentation_amd import detect, merge
    import datetime
    now = datetime.datetime(2020, 1, 2, 3, 4, 5)
    rows = detect.csv_rows([[100, 200, 300, 400, 0.9], [110, 190, 310, 410, 0.8]], 8000, 16000, 8.0, "site", "PAS 001", "PAS-001.ndpi", now)
    rows += detect.csv_rows([[0, 0, 50, 50, 0.95]], 0, 0, 8.0, "site", "PAS 002", "PAS-002.ndpi", now)
    det = tmp_path / "det.csv"
    det.write_text("".join(rows))
    groups = merge.read_detections_csv(str(det))
    assert [g[2] for g in groups] == ["PAS-001.ndpi", "PAS-002.ndpi"] and groups[0][3][0] == [8800.0, 17600.0, 10400.0, 19200.0, 0.9]
    out = tmp_path / "merged.csv"
    merge.merge_csv(str(det), str(out), lambda s, f: (0.2277, 0.2277), 0.35)
    text = out.read_text().splitlines()
    assert text[0] == 'site,PAS 001,"PAS-001.ndpi",8800,17520,10480,19280,0.9'      # the two windows' boxes merged
    assert text[1] == 'site,PAS 002,"PAS-002.ndpi",0,0,400,400,0.95'
    lists, order = merge.read_merged_csv(str(out))
    assert order == ["PAS001", "PAS002"] and lists["PAS001"][0] == [8800, 17520, 10480, 19280, 0.9]
    assert merge.crop_name(lists["PAS001"][0]) == "xmin1100_ymin2190_xmax1310_ymax2410"


def test_contours_and_polygons():
    """host contour tracer + polygon simplifier: geometric invariants (cv2 parity is unpinned)"""
    from glomeruli_segmentation_amd import contours
    img = np.zeros((40, 60), np.uint8)
    img[5:25, 10:40] = 1            # a rectangle ...
    img[10:20, 20:30] = 0           # ... with a hole
    img[30:33, 50:53] = 1           # and a small blob
    cs = contours.find_contours(img)
    assert len(cs) == 3                                               # outer, hole, blob (RETR_LIST keeps holes)
    outer = [c for c in cs if c[:, 0, 0].min() == 10 and c[:, 0, 0].max() == 39][0]
    assert sorted(map(tuple, outer[:, 0, :].tolist())) == [(10, 5), (10, 24), (39, 5), (39, 24)]   # CHAIN_APPROX_SIMPLE: corners only
    hole = [c for c in cs if c[:, 0, 0].min() == 19][0]               # a hole border runs on the foreground ring around it
    assert hole[:, 0, 0].max() == 30 and hole[:, 0, 1].min() == 9 and hole[:, 0, 1].max() == 20
    full = contours.find_contours(img, simple=False)
    assert max(len(c) for c in full) == 2 * (30 + 20) - 4            # every border pixel of the rectangle once
    assert abs(contours.arc_length(outer) - 2 * (29 + 19)) < 1e-9
    # a disc: every border point lies on the disc's boundary, the polygon stays within epsilon of it
    yy, xx = np.mgrid[0:200, 0:200]
    disc = ((yy - 100) ** 2 + (xx - 100) ** 2 <= 70 ** 2).astype(np.uint8)
    c = contours.find_contours(disc, simple=False)[0][:, 0, :]
    r = np.hypot(c[:, 0] - 100, c[:, 1] - 100)
    assert r.min() > 68.5 and r.max() <= 70.0 and len(c) > 350
    assert (np.abs(np.diff(np.vstack([c, c[:1]]), axis=0)).max(axis=1) == 1).all()      # 8-connected closed chain
    eps = 0.003 * contours.arc_length(c)
    poly = contours.approx_poly(c, eps)[:, 0, :]
    assert 8 <= len(poly) < len(c) // 4
    seg_a, seg_b = poly, np.roll(poly, -1, axis=0)
    # known answer worked by hand through OpenCV 4.3's approxPolyDP_ (the algorithm contours.cpp restates): a 10 x 2 rectangle
    # with mid-points on its long sides, given from a mid-point.  Three farthest-point hops from P0 = (5,0) go to (10,2), to
    # (0,0), to (10,2): the cut is (0,0) | (10,2), the mid-points have distance 0 from their chords and drop out, the
    # clean-up pass keeps the four corners -- and the polygon STARTS at the cut (0,0), not at the first input point
    rect = np.array([[5, 0], [10, 0], [10, 2], [5, 2], [0, 2], [0, 0]], dtype=np.int32)
    assert contours.approx_poly(rect, 0.5)[:, 0, :].tolist() == [[0, 0], [10, 0], [10, 2], [0, 2]]
    assert contours.approx_poly(rect, 100.0)[:, 0, :].tolist() == [[0, 0]]             # everything within epsilon: one point (:le_eps)
    for p in c[::7]:                                                  # each original point is within (1 + sqrt(1/2)) eps of the polygon
        d = []
        for a, b in zip(seg_a, seg_b):
            ab, ap = (b - a).astype(float), (p - a).astype(float)
            t = np.clip(ap @ ab / max(ab @ ab, 1e-12), 0, 1)
            d.append(np.hypot(*(ap - t * ab)))
        assert min(d) <= (1.0 + 0.5 ** 0.5) * eps + 1e-9                 # RDP bound + what the clean-up pass may remove
    gy, gx = np.mgrid[0:300, 0:300]
    cm = np.zeros((300, 300), np.uint8)
    cm[(gy - 150) ** 2 + (gx - 150) ** 2 <= 120 ** 2] = 1
    cm[(gy - 160) ** 2 + (gx - 170) ** 2 <= 35 ** 2] = 3          # round blob: many direction changes
    cm[20:24, 20:24] = 2                                            # 4-corner speck: below o_min_points, dropped as noise
    d = contours.labelme_dict(cm, "x.PNG")
    labels = [s["label"] for s in d["shapes"]]
    assert labels.count("glomerulus") == 1 and labels.count("sclerosis") == 1 and "crescent" not in labels
    assert d["imagePath"] == "x.PNG" and all(len(s["points"]) >= 3 for s in d["shapes"])


def test_contours_against_scipy_topology():
    """the border follower pinned on what scipy.ndimage (not ours) says about the same images: one outer border per
    8-connected component, one hole border per enclosed 4-connected background component, and the traced points are
    exactly the foreground pixels with a background 4-neighbour (Suzuki-Abe's border points, the set
    cv2.findContours(RETR_LIST, CHAIN_APPROX_NONE) visits -- boundary_extractor.py:33-36)"""
    from scipy import ndimage as ndi
    from glomeruli_segmentation_amd import contours
    rng = np.random.default_rng(3)
    s8, s4 = np.ones((3, 3), int), ndi.generate_binary_structure(2, 1)
    for trial in range(8):
        field = ndi.gaussian_filter(rng.standard_normal((120, 160)), 1 + trial)
        img = (field > 0.02 / (1 + trial)).astype(np.uint8)
        cs = contours.find_contours(img, simple=False)
        _, ncomp = ndi.label(img, structure=s8)
        bl, nb = ndi.label(1 - img, structure=s4)
        frame = (set(bl[0, :]) | set(bl[-1, :]) | set(bl[:, 0]) | set(bl[:, -1])) - {0}
        assert len(cs) == ncomp + (nb - len(frame)), trial
        traced = set()
        for c in cs:
            traced |= set(map(tuple, c[:, 0, :].tolist()))
        pad = np.pad(img, 1)
        border = (img == 1) & ((pad[:-2, 1:-1] == 0) | (pad[2:, 1:-1] == 0) | (pad[1:-1, :-2] == 0) | (pad[1:-1, 2:] == 0))
        assert traced == set(zip(*np.nonzero(border)[::-1])), trial
        # CHAIN_APPROX_SIMPLE keeps a subset of the same points, in the same number of contours
        simple = contours.find_contours(img, simple=True)
        assert len(simple) == len(cs)
        assert all(set(map(tuple, a[:, 0, :].tolist())) <= set(map(tuple, b[:, 0, :].tolist())) for a, b in zip(simple, cs))


def test_cabi_argument_and_device_errors_without_gpu(sd1):
    """status codes + gs_last_error through the C ABI; no compute is attempted (this box has no GPU, and on a GPU
    box the same calls stop at argument validation)"""
    import ctypes
    from glomeruli_segmentation_amd import _lib
    from glomeruli_segmentation_amd.engine import pack_state_dict
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.gs_espnet_create(None, None, 0, 5, 2, 8, 0, ctypes.byref(h)) == 1          # GS_ERR_INVALID
    assert b"null" in lib.gs_last_error()
    blob, table = pack_state_dict(sd1)
    for bad in (1, 21, 0, -3):                                                              # 2 <= classes <= GS_MAX_CLASSES
        rc = lib.gs_espnet_create(blob.ctypes.data_as(ctypes.c_void_p), table, len(table), bad, 2, 8, 0, ctypes.byref(h))
        assert rc == 4 and b"classes must be 2..20" in lib.gs_last_error()                  # GS_ERR_UNSUPPORTED
    assert lib.gs_conv2d_nhwc(None, 1, 8, 8, 3, None, 3, 3, 4, None, 1, 1, 0, None, None) == 1
    assert lib.gs_nms(None, None, 0, ctypes.c_float(0.5), ctypes.c_float(0.0), 10, None, None, None) == 1
    assert lib.gs_crop_preprocess(None, 4, 4, None, None, 8, 8, None, None) == 1
    # the round-3 entries validate before they touch a device: null model list, bad network size, nothing to compute
    f3 = (ctypes.c_float * 3)(1.0, 1.0, 1.0)
    assert lib.gs_espnet_segment_crops_host(None, 1, None, None, None, 1, f3, f3, 512, 1024, 32, None, None, None, None, None, None, None) == 1
    assert b"null" in lib.gs_last_error()
    assert lib.gs_espnet_segment_crops(None, 0, None, None, 1, f3, f3, 512, 1024, None, None, None, None, None) == 1
    assert lib.gs_espnet_ensemble_segment_crops(None, 0, None, None, 1, f3, f3, 512, 1024, None, None, None, None, None) == 1
    assert lib.gs_detector_detect_host(None, None, 1, 64, 64, 1, None, None, None, None) == 1
    assert lib.gs_abi_version() == _lib.ABI_VERSION and lib.gs_build_flags() == 0          # the product build: no GS_DIAG
    import torch
    if not torch.cuda.is_available():
        rc = lib.gs_espnet_create(blob.ctypes.data_as(ctypes.c_void_p), table, len(table), 5, 2, 8, 0, ctypes.byref(h))
        assert rc in (2, 5), rc                                                              # GS_ERR_HIP / GS_ERR_NODEVICE: loud, no fallback
        assert lib.gs_device_fault_check() == 2 and lib.gs_last_error()                      # no device to synchronise: GS_ERR_HIP, not "no fault"
        from glomeruli_segmentation_amd.engine import EspnetEngine
        with pytest.raises(RuntimeError):
            EspnetEngine(sd1)


# ------------------------------------------------------------------------------------------------------------------
# launcher: fail fast, CPU placement
def test_a_failing_rank_takes_the_job_down_quickly():
    """`bench.py --gpus 2 --dry-run --fail-rank 1`: rank 1 exits with code 3 after the rendezvous while rank 0 sits in a
    collective; the parent terminates rank 0, reports rank 1's stderr tail and returns non-zero within seconds"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run",
                        "--fail-rank", "1"], env=env, capture_output=True, text=True, timeout=120)
    el = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    assert el < 10.0, el
    assert "rank 1 of 2 exited with code 3" in p.stderr and "fails on purpose" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]      # no result line from a broken job


def test_a_rank_that_raises_inside_a_cli_ends_the_job(tmp_path):
    """A sharded CLI whose rank 1 raises in its window loop while rank 0 goes on to the row gather: rank 1 must leave non-zero
    at once (no barrier on the failure path, shard.abort_rank) so that spawn_ranks can end rank 0 -- with the barrier in a
    `finally:` rank 1 sat in a collective its peer never reached, until the backend's timeout."""
    import io
    import time
    from glomeruli_segmentation_amd import launch
    data_dir, tl, _ = _png_slide_tree(tmp_path, W=8192, H=4096)
    helper = os.path.join(REPO, "tests", "helpers", "detect_stub_rank.py")
    argv = ["--target_list", tl, "--data_dir", data_dir, "--staining", "OPT_PAS", "--output_dir", str(tmp_path / "out"),
            "--window_size", "500", "--overlap_ratio", "0.1", "--batch", "4"]
    saved = {k: os.environ.get(k) for k in ("GLOMSEG_DIST_BACKEND", "GS_TEST_FAIL_RANK", "GS_TEST_FAIL_HOW", "MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        for k in ("MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK", "GS_TEST_FAIL_HOW"):
            os.environ.pop(k, None)
        os.environ["GLOMSEG_DIST_BACKEND"] = "gloo"
        os.environ["GS_TEST_FAIL_RANK"] = "1"
        out, err = io.StringIO(), io.StringIO()
        t0 = time.time()
        rc = launch.spawn_ranks(helper, argv, 2, out=out, err=err)
        el = time.time() - t0
        assert rc == 1 and el < 60.0, (rc, el, err.getvalue()[-800:])
        assert "rank 1 of 2 exited with code 1" in err.getvalue() and "fails on purpose" in err.getvalue()
        # shard.abort_rank's contract: ANY exit of one rank inside the sharded region fails the job -- SystemExit(0) and
        # SystemExit("message") leave with status 1 (the peers wait in a collective this rank never joins), a non-zero
        # integer code is kept
        for how, want in (("exit0", 1), ("exit_message", 1), ("exit7", 7)):
            os.environ["GS_TEST_FAIL_HOW"] = how
            os.environ.pop("MASTER_PORT", None)
            out, err = io.StringIO(), io.StringIO()
            t0 = time.time()
            rc = launch.spawn_ranks(helper, argv, 2, out=out, err=err)
            assert rc == want and time.time() - t0 < 60.0, (how, rc, err.getvalue()[-800:])
            assert "rank 1 of 2 exited with code %d" % want in err.getvalue(), (how, err.getvalue()[-800:])
        os.environ.pop("GS_TEST_FAIL_HOW", None)
        # and the same command line with nobody failing writes the one CSV
        os.environ["GS_TEST_FAIL_RANK"] = "none"
        os.environ.pop("MASTER_PORT", None)
        rc = launch.spawn_ranks(helper, argv, 2, out=io.StringIO(), err=io.StringIO())
        assert rc == 0 and os.path.isfile(tmp_path / "out" / "OPT_PAS_GlomusList.csv")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_rank_cpu_placement_follows_the_gpu_numa_node(tmp_path):
    """launch.rank_cpus on a fake sysfs: four GPUs, two per NUMA node; ranks share their node's allowed CPUs evenly"""
    from glomeruli_segmentation_amd import launch
    sysfs = tmp_path / "sys"
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 16\nsimd_count 0\n")
    for k, (bus, numa) in enumerate([(0x05, 0), (0x15, 0), (0x85, 1), (0x95, 1)]):
        d = nodes / str(k + 1)
        d.mkdir()
        d.joinpath("properties").write_text("simd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        pci = sysfs / "bus" / "pci" / "devices" / ("0000:%02x:00.0" % bus)
        pci.mkdir(parents=True)
        pci.joinpath("numa_node").write_text("%d\n" % numa)
    for n, cl in ((0, "0-7,16-23"), (1, "8-15,24-31")):
        d = sysfs / "devices" / "system" / "node" / ("node%d" % n)
        d.mkdir(parents=True)
        d.joinpath("cpulist").write_text(cl + "\n")
    assert launch.gpu_numa_nodes(str(sysfs)) == [0, 0, 1, 1]
    allowed = set(range(32))
    got = [launch.rank_cpus(r, 4, allowed, str(sysfs), {}) for r in range(4)]
    assert got[0] == [0, 1, 2, 3, 4, 5, 6, 7] and got[1] == [16, 17, 18, 19, 20, 21, 22, 23]
    assert got[2] == [8, 9, 10, 11, 12, 13, 14, 15] and got[3] == [24, 25, 26, 27, 28, 29, 30, 31]
    # a cgroup that allows only part of a node; no topology at all -> even split of what is allowed
    assert launch.rank_cpus(1, 4, set(range(0, 20)), str(sysfs), {}) == [6, 7, 16, 17, 18, 19]      # node 0 within the cgroup: 0-7,16-19
    assert [launch.rank_cpus(r, 2, set(range(6)), str(tmp_path / "none")) for r in range(2)] == [[0, 1, 2], [3, 4, 5]]
    assert launch.rank_cpus(0, 1, {3, 4}, str(sysfs)) == [3, 4]
    # a visible-device list re-maps local ranks to physical GPUs: ranks 0, 1 on physical GPUs 2, 3 (both on node 1)
    assert launch.visible_gpu_indices(4, {"HIP_VISIBLE_DEVICES": "2,3"}) == [2, 3]
    assert launch.visible_gpu_indices(4, {"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "1,2"}) == [2, 3]
    assert launch.visible_gpu_indices(4, {"HIP_VISIBLE_DEVICES": "GPU-abc"}) == [0, 1, 2, 3]
    assert launch.rank_cpus(0, 2, allowed, str(sysfs), {"HIP_VISIBLE_DEVICES": "2,3"}) == [8, 9, 10, 11, 12, 13, 14, 15]


# ------------------------------------------------------------------------------------------------------------------
# detector / merge command lines (detect_glomus_test.py:385-456,93-159,196-234; merge_overlaped_glomus.py:385-389)
def _png_slide_tree(tmp_path, W=53248, H=23040, ds=8.0, mpp=0.2277):
    """data_dir/<staining dir>/<specimen>/<file>.PNG = a slide at 1/ds, plus its target-list line"""
    from PIL import Image
    from glomeruli_segmentation_amd.synth import synth_tile
    data_dir = tmp_path / "kidney" / "site_a"
    sdir = data_dir / "02_PAS" / "H16-0001"
    sdir.mkdir(parents=True)
    img = synth_tile(9, int(H / ds), int(W / ds), blobs=8)[:, :, ::-1]
    Image.fromarray(np.ascontiguousarray(img)).save(sdir / "H16-0001_PAS.PNG")
    tl = tmp_path / "target_list.txt"
    tl.write_text("#H16-0000/H16-0000_PAS,1,1,40,8,0.2,0.2\nH16-0001/H16-0001_PAS,%d,%d,40,%g,%g,%g\nH16-0404/missing,1,1,40,8,0.2,0.2\n"
                  % (W, H, ds, mpp, mpp))
    return str(data_dir) + "/", str(tl), np.ascontiguousarray(img)


def test_detect_cli_png_branch_with_a_stub_detector(tmp_path):
    """parser + target list + window walk over a PNG slide + CSV / log files, with a stub behind the detect_box contract"""
    from glomeruli_segmentation_amd import detect
    data_dir, tl, img = _png_slide_tree(tmp_path)
    seen = []

    def stub(ims):
        ims = np.asarray(ims)
        seen.append(ims.shape)
        n = len(ims)
        b = np.zeros((n, 2, 4), np.float32)
        b[:, 0] = [0.25, 0.5, 0.5, 0.75]                      # ymin, xmin, ymax, xmax
        b[:, 1] = [0.0, 0.0, 0.1, 0.1]
        s = np.tile(np.array([[0.9, 0.3]], np.float32), (n, 1))
        return b, s, np.ones_like(s), np.full((n,), 2.0, np.float32)

    out_dir = tmp_path / "out" / "detect"
    rc = detect.main(["--target_list", tl, "--data_dir", data_dir, "--staining", "OPT_PAS", "--output_dir", str(out_dir),
                      "--window_size", "2000", "--overlap_ratio", "0.1", "--conf_threshold", "0.6", "--batch", "1"], detector=stub)
    assert rc == 0
    plan = detect.plan_windows(53248, 23040, 0.2277, 0.2277, 8.0, 2000, 0.1, from_image=True)
    assert (plan.x_split_times, plan.y_split_times, plan.window_x) == (7, 3, 1098)
    assert len(seen) == 21 and all(sh == (1, 1098, 1098, 3) for sh in seen)
    rows = open(out_dir / "OPT_PAS_GlomusList.csv").read().splitlines()
    assert len(rows) == 21                                   # one box per window passes the threshold
    f = rows[0].split(",")
    assert f[:4] == ['"site_a"', '"H16-0001"', '"H16-0001_PAS.PNG"', 'new']
    # window (0,0): x1 = 0*8 + int(1098*0.5)*8, y1 = int(1098*0.25)*8 (:234, :319-325)
    assert [float(v) for v in f[5:9]] == [549 * 8.0, 274 * 8.0, 823 * 8.0, 549 * 8.0] and f[9] == "0.9"
    last = rows[-1].split(",")                                # window (6,2): origin = stride * index, lifted by the downsample
    assert float(last[5]) == plan.step_x * 6 * 8.0 + 549 * 8.0 and float(last[6]) == plan.step_y * 2 * 8.0 + 274 * 8.0
    log = open(out_dir / "OPT_PAS_GlomusList_log.csv").read().splitlines()
    assert log[0] == "file,time" and len(log) == 2 and log[1].startswith('"H16-0001_PAS",')
    # windows past the right / bottom edge are padded with black, as PIL's crop pads them
    with pytest.raises(ValueError):
        detect.staining_type("OPT_XYZ")
    assert detect.staining_dir("OPT_HE") == "" and detect.staining_dir("OPT_PAM") == "03_PAM"
    # defaults (500 um, 0.5) when --window_size is absent (:52-54)
    seen.clear()
    detect.main(["--target_list", tl, "--data_dir", data_dir, "--staining", "OPT_PAS", "--output_dir", str(out_dir), "--output_file_ext",
                 "_default", "--batch", "64"], detector=stub)
    p2 = detect.plan_windows(53248, 23040, 0.2277, 0.2277, 8.0, None, None, from_image=True)
    assert sum(sh[0] for sh in seen) == p2.x_split_times * p2.y_split_times == 49 * 21
    assert os.path.isfile(out_dir / "OPT_PAS_default.csv")
    # a .pb graph is refused with an explanation; a missing file likewise
    (tmp_path / "m").mkdir()
    (tmp_path / "m" / "frozen_inference_graph.pb").write_bytes(b"x")
    with pytest.raises(ValueError, match="frozen graph"):
        detect.load_detector(str(tmp_path / "m"), "frozen_inference_graph.pb")
    with pytest.raises(FileNotFoundError):
        detect.load_detector(str(tmp_path / "nope"), "frozen_inference_graph.pb")


def test_merge_cli(tmp_path):
    """merge_overlaped_glomus.py's command line: detections CSV + target list -> merged CSV + log"""
    from glomeruli_segmentation_amd import merge
    z = load_golden("merge.npz")
    dets = z["in_0"]
    det_csv = tmp_path / "OPT_PAS_GlomusList.csv"
    with open(det_csv, "w") as f:
        for d in dets:
            f.write('"site_a","H16-0001","H16-0001_PAS.PNG",new,2020-01-01T00:00:00,%s,%s,%s,%s,%s\n' % tuple(str(float(v)) for v in d[:5]))
    tl = tmp_path / "tl.txt"
    tl.write_text("H16-0001/H16-0001_PAS,53248,23040,40,8,0.2277,0.2277\n")
    rc = merge.main(["--staining", "OPT_PAS", "--target_list", str(tl), "--detected_list", str(det_csv), "--output_dir", str(tmp_path),
                     "--output_file_ext", "test", "--conf_threshold", "0.2", "--overlap_threshold", "0.35"])
    assert rc == 0
    rows = open(tmp_path / "OPT_PAS_GlomusMergedList_test.csv").read().splitlines()
    want = merge.merge_detections([list(map(float, d[:5])) for d in dets], 0.2277, 0.2277, 0.35, 0.2)
    assert rows == [r.rstrip("\n") for r in merge.merged_csv_rows("site_a", "H16-0001", "H16-0001_PAS.PNG", want)]
    log = open(tmp_path / "OPT_PAS_GlomusMergedList_test_log.csv").read().splitlines()
    assert len(log) == 1 and log[0].startswith('"H16-0001_PAS.PNG",')
    boxes, order = merge.read_merged_csv(tmp_path / "OPT_PAS_GlomusMergedList_test.csv")
    assert order == ["H16-0001"] and len(boxes["H16-0001"]) == len(want)
    # a non-PNG slide without OpenSlide and without a target-list line fails loudly
    with open(det_csv, "w") as f:
        f.write('"site_a","H16-0002","H16-0002_PAS.ndpi",new,2020-01-01T00:00:00,0.0,0.0,100.0,100.0,0.9\n')
    with pytest.raises(RuntimeError, match="openslide"):
        merge.main(["--staining", "OPT_PAS", "--target_list", str(tl), "--detected_list", str(det_csv), "--output_dir", str(tmp_path),
                    "--overlap_threshold", "0.35"])


# ------------------------------------------------------------------------------------------------------------------
# the reference's documented command lines parse as they stand (example/README.md:25-133, README.md:226-281)
def _argv(text):
    """a README command block -> argv: the words after the script name, continuation backslashes dropped"""
    import shlex
    words = shlex.split(text.replace("\\\n", " "))
    return words[2:]                      # drop `python <script>`


def test_crop_cli_writes_the_reference_file_contract(tmp_path):
    """python -m glomeruli_segmentation_amd.crop = make_seg_data.py's no-ground-truth branch (:347-361) over a PNG slide:
    one RGBA PNG per merged box under org_image/<slide>/, named by its level-0 coordinates / 8, of the box's size"""
    from PIL import Image
    from glomeruli_segmentation_amd import crop, merge
    data_dir, tl, img = _png_slide_tree(tmp_path, W=8000, H=4000)
    merged = tmp_path / "OPT_PAS_GlomusMergedList_t.csv"
    boxes = [[800, 160, 1500, 900], [3203, 1001, 4100, 1999], [7000, 3000, 8000, 4000]]
    merged.write_text("".join('site_a,H16-0001,"H16-0001_PAS.PNG",%d,%d,%d,%d,0.9\n' % tuple(b) for b in boxes))
    out = tmp_path / "seg_data"
    rc = crop.main(["--staining=OPT_PAS", "--target_list=" + tl, "--merged_detection_result_csv=" + str(merged),
                    "--wsi_dir=" + os.path.join(data_dir, "02_PAS"), "--output_dir=" + str(out)])
    assert rc == 0
    files = sorted(os.listdir(out / "org_image" / "H16-0001"))
    assert files == sorted(merge.crop_name(b) + ".PNG" for b in boxes)
    assert "xmin400_ymin125_xmax512_ymax249.PNG" in files                      # int(3203 / 8) = 400, int(1001 / 8) = 125, ...
    for b in boxes:
        with Image.open(out / "org_image" / "H16-0001" / (merge.crop_name(b) + ".PNG")) as im:
            a = np.asarray(im)
        assert a.shape == (b[3] - b[1], b[2] - b[0], 4) and (a[:, :, 3] == 255).all()
        ys, xs = (b[1] + np.arange(b[3] - b[1])) // 8, (b[0] + np.arange(b[2] - b[0])) // 8
        assert (a[:, :, :3] == img[np.minimum(ys, img.shape[0] - 1)][:, np.minimum(xs, img.shape[1] - 1)]).all()
    # the ground-truth branch (both the JSON and the XML directory given, :388) is refused with an explanation
    assert crop.main(["--staining=OPT_PAS", "--target_list=" + tl, "--merged_detection_result_csv=" + str(merged), "--wsi_dir=w",
                      "--segmentation_gt_json_dir=a", "--object_detection_gt_xml_dir=b"]) == 2
    # a slide without metadata in the target list cannot be read without OpenSlide: a clear error, not a silent skip
    (tmp_path / "tl2.txt").write_text("H16-0001/H16-0001_PAS\n")
    with pytest.raises(RuntimeError):
        crop.main(["--staining=OPT_PAS", "--target_list=" + str(tmp_path / "tl2.txt"), "--merged_detection_result_csv=" + str(merged),
                   "--wsi_dir=" + os.path.join(data_dir, "02_PAS"), "--output_dir=" + str(out)])


def test_reference_readme_command_lines_parse():
    from glomeruli_segmentation_amd import composite, crop, detect, merge, segment
    out = "/workspace/output"
    # --- example/README.md:26-37, detect_glomus_test.py
    a = detect.build_parser().parse_args(_argv("""python /opt/glomeruli_detection/detect_glomus_test.py \
        --model=/workspace/fold_1 \
        --target_list=/opt/ESPNet/example/opt_pas_test_list.txt \
        --data_dir=/opt/ESPNet/example/data \
        --staining=OPT_PAS \
        --output_dir=%s \
        --output_file_ext=_test1 \
        --window_size=2000 \
        --overlap_ratio=0.1 \
        --conf_threshold=0.2 \
        --model_name=frozen_inference_graph.pb""" % out))
    assert (a.model, a.data_category, a.output_file_ext, a.window_size, a.overlap_ratio, a.conf_threshold, a.model_name) == \
        ("/workspace/fold_1", "OPT_PAS", "_test1", 2000, 0.1, 0.2, "frozen_inference_graph.pb")
    d = detect.build_parser().parse_args([])          # defaults of detect_glomus_test.py:390-403
    assert (d.data_category, d.output_dir, d.output_file_ext, d.window_size, d.overlap_ratio, d.conf_threshold) == \
        ("OPT_PAM", "./output", "_GlomusList", None, None, 0.6)
    # --- example/README.md:40-49, merge_overlaped_glomus.py
    a = merge.build_parser().parse_args(_argv("""python /opt/glomeruli_detection/merge_overlaped_glomus.py \
        --target_list=/opt/ESPNet/example/opt_pas_test_list.txt \
        --detected_list=/workspace/output/OPT_PAS_test1.csv \
        --data_dir=/opt/ESPNet/example/data \
        --staining=OPT_PAS \
        --output_dir=/workspace/output \
        --output_file_ext=test1 \
        --conf_threshold=0.9 \
        --overlap_threshold=0.35"""))
    assert (a.input_file, a.annotation_dir, a.training_type, a.conf_threshold, a.overlap_threshold) == \
        ("/workspace/output/OPT_PAS_test1.csv", "/opt/ESPNet/example/data", "test1", 0.9, 0.35)
    # --- example/README.md:52-71 (both forms), make_seg_data.py
    for extra in ("--segmentation_gt_json_dir=/opt/ESPNet/example/data/seg_annotation \\\n--object_detection_gt_xml_dir=/opt/ESPNet/example/data \\\n"
                  "--segmentation_gt_png_dir=/opt/ESPNet/example/data/label \\\n", ""):
        a = crop.build_parser().parse_args(_argv("""python /opt/glomeruli_detection/make_seg_data.py \
            --staining=OPT_PAS \
            --target_list=/opt/ESPNet/example/opt_pas_test_list.txt \
            --merged_detection_result_csv=%s/OPT_PAS_GlomusMergedList_test1.csv \
            %s--wsi_dir=/opt/ESPNet/example/data/02_PAS \
            --output_dir=%s/seg_data""" % (out, extra, out)))
        assert a.wsi_dir.endswith("02_PAS") and a.output_dir.endswith("seg_data") and (a.gt_png_dir is not None) == bool(extra)
    d = crop.build_parser().parse_args(["--staining", "OPT_PAS", "--merged_detection_result_csv", "m", "--target_list", "t", "--wsi_dir", "w"])
    assert (d.iou_threshold, d.output_dir, d.start, d.end, d.no_save) == (0.01, "./output/seg_data", 0, 0, False)    # :372-379
    # --- example/README.md:75-104 (both forms) and README.md:226-239, VisualizeResults_iou.py
    for label in ("--label_data_dir=%s/seg_data/label/all \\\n" % out, ""):
        a = segment.build_parser().parse_args(_argv("""python /opt/ESPNet/test/VisualizeResults_iou.py \
            --classes=5 \
            --rgb_data_dir=%s/seg_data/org_image \
            %s--savedir=%s/seg_data_pred \
            --weights=/opt/ESPNet/models/espnet_fold1.pth \
            --gpu_id=0 \
            --decoder \
            --img_extn=PNG \
            --colored \
            --overlay \
            --mean 204.60071 170.19359 199.57469 \
            --std 20.61257 42.92207 28.401505""" % (out, label, out)))
        assert a.classes == 5 and a.gpu_id == 0 and a.decoder and a.colored and a.overlay and not a.cityFormat
        assert a.mean == ["204.60071", "170.19359", "199.57469"] and a.std == ["20.61257", "42.92207", "28.401505"]
        assert (a.label_data_dir is not None) == bool(label)
    a = segment.build_parser().parse_args(_argv("""python /opt/ESPNet/test/VisualizeResults_iou.py \
        --classes 5 --rgb_data_dir d --label_data_dir l --savedir s --weights /models/espnet_fold1.pth --gpu_id 0 --img_extn PNG \
        --mean 204.60071 170.19359 199.57469 --std 20.61257 42.92207 28.401505 --decoder --colored --overlay --cityFormat"""))
    assert a.cityFormat and a.img_extn == "PNG"
    d = segment.build_parser().parse_args(["--rgb_data_dir", "d", "--weights", "w", "--mean", "1", "2", "3", "--std", "1", "2", "3"])
    assert (d.inWidth, d.inHeight, d.scaleIn, d.modelType, d.savedir, d.gpu_id, d.p, d.q, d.classes, d.img_extn) == \
        (1024, 512, 1, 1, "./results", -1, 2, 8, 5, "PNG")          # VisualizeResults_iou.py:295-315
    # --- example/README.md:108-133, eval_wsi_segmentation.py (with and without the ground-truth directories)
    gt = ("--segmentation_gt_json_dir=/opt/ESPNet/example/data/seg_annotation \\\n"
          "--object_detection_gt_xml_dir=/opt/ESPNet/example/data \\\n--segmentation_gt_png_dir=/opt/ESPNet/example/data/label \\\n")
    for extra in (gt, ""):
        a = composite.build_parser().parse_args(_argv("""python /opt/ESPNet/test/eval_wsi_segmentation.py \
            --staining=OPT_PAS \
            --target_list=/opt/ESPNet/example/opt_pas_test_list.txt \
            --merged_detection_result_csv=%s/OPT_PAS_GlomusMergedList_test1.csv \
            %s--wsi_dir=/opt/ESPNet/example/data/02_PAS \
            --output_file=%s/seg_data_pred/seg_data_output.tsv \
            --segmentation_pred_json_dir=%s/seg_data_pred \
            --window_size=2400 \
            --output_dir=%s/seg_data_pred""" % (out, extra, out, out, out)))
        assert a.output_file.endswith("seg_data_output.tsv") and a.window_size == 2400 and a.staining == "OPT_PAS"
        assert (a.seg_gt_json_dir is not None) == bool(extra)
    d = composite.build_parser().parse_args(["--staining", "OPT_PAS", "--merged_detection_result_csv", "m", "--target_list", "t",
                                             "--wsi_dir", "w", "--segmentation_pred_json_dir", "j", "--start", "3", "--end", "9",
                                             "--iou_threshold", "0.5"])
    assert (d.iou_threshold, d.start, d.end, d.output_file, d.output_dir, d.window_size, d.classes, d.no_save) == \
        (0.5, 3, 9, "./output/seg_data_pred/seg_data_output.tsv", "./output/seg_data_pred", 2400, 5, False)   # :412-420
    # the evaluation branch (all three ground-truth directories, :427) is refused with an explanation, not by argparse
    rc = composite.main(["--staining", "OPT_PAS", "--merged_detection_result_csv", "m", "--target_list", "t", "--wsi_dir", "w",
                         "--segmentation_pred_json_dir", "j", "--segmentation_gt_json_dir", "a", "--object_detection_gt_xml_dir", "b",
                         "--segmentation_gt_png_dir", "c"])
    assert rc == 2
    # the reference's default device (--gpu_id -1 = CPU, VisualizeResults_iou.py:252-255,302) is refused likewise: this build
    # has no CPU product path
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".pth") as f:
        assert segment.main(["--rgb_data_dir", "d", "--weights", f.name, "--mean", "1", "2", "3", "--std", "1", "2", "3"]) == 2


def test_segment_cli_overlapped_host_work_writes_the_same_bytes(tmp_path, monkeypatch):
    """segment.evaluate with a decode-ahead / write-behind pool against its serial loop (--workers 0): every output file byte
    for byte, rows in list order -- with labels (accuracy rows, combined images, the summed confusion matrix) and without.
    The GPU pass is replaced by a deterministic stand-in: this is about the host pipeline around it."""
    import filecmp
    from PIL import Image
    from glomeruli_segmentation_amd import segment
    from glomeruli_segmentation_amd.synth import synth_tile

    def fake_segment_images(engine, images, mean, std, width, height, batch, want_net_maps=False):
        outs, nets = [], []
        for im in images:
            cm = (im[:, :, 0].astype(np.int32) // 52).astype(np.uint8) % 5
            ys = (np.arange(height) * im.shape[0] // height)
            xs = (np.arange(width) * im.shape[1] // width)
            outs.append(cm)
            nets.append(np.ascontiguousarray(cm[ys][:, xs]))
        return (outs, nets) if want_net_maps else outs
    monkeypatch.setattr(segment, "segment_images", fake_segment_images)
    rng = np.random.default_rng(3)
    rgb, lab = tmp_path / "rgb", tmp_path / "lab"
    n = 23
    for k in range(n):
        patient = "P%d" % (k % 3)
        (rgb / patient).mkdir(parents=True, exist_ok=True)
        (lab / patient).mkdir(parents=True, exist_ok=True)
        tile = synth_tile(k, 32, 64, blobs=3)
        Image.fromarray(np.ascontiguousarray(tile[:, :, ::-1])).save(rgb / patient / ("xmin%d_ymin0_xmax9_ymax9.PNG" % k))
        Image.fromarray(rng.integers(0, 5, (32, 64), dtype=np.uint8)).save(lab / patient / ("xmin%d_ymin0_xmax9_ymax9.PNG" % k))
    trees = {}
    for tag, workers, with_labels in (("serial_l", 0, True), ("pool_l", 5, True), ("serial", 0, False), ("pool", 3, False)):
        out = tmp_path / tag
        argv = ["--rgb_data_dir", str(rgb), "--savedir", str(out), "--weights", "unused", "--mean", "1", "2", "3", "--std", "1", "2", "3",
                "--inWidth", "64", "--inHeight", "32", "--batch", "4", "--colored", "--overlay", "--cityFormat", "--workers", str(workers)]
        if with_labels:
            argv += ["--label_data_dir", str(lab)]
        args = segment.build_parser().parse_args(argv)
        rgb_list = sorted(__import__("glob").glob(str(rgb) + "/*/*.PNG"))
        label_list = sorted(__import__("glob").glob(str(lab) + "/*/*.PNG")) if with_labels else [None] * n
        # (a stand-in engine of the ESPNet-C kind: segment_batch then takes its maps from segment_images -- the stand-in above --
        # and the counts / overlays from the host arithmetic the GPU pass is tested against in the gpu suite)
        import types
        segment.evaluate(args, types.SimpleNamespace(encoder_only=True, classes=5, device=None), rgb_list, label_list)
        files = sorted(os.path.relpath(os.path.join(d, f), out) for d, _, fs in os.walk(out) for f in fs)
        trees[tag] = (out, files)
    for a, b in (("serial_l", "pool_l"), ("serial", "pool")):
        (oa, fa), (ob, fb) = trees[a], trees[b]
        assert fa == fb and len(fa) > 4 * n
        for f in fa:
            assert filecmp.cmp(os.path.join(oa, f), os.path.join(ob, f), shallow=False), f
    rows = open(trees["pool_l"][0] / "summary_pixel.csv").read().splitlines()[1:]
    assert [r.split(",")[1] for r in rows] == [os.path.basename(p).replace("PNG", "png") for p in rgb_list]      # list order
    assert len(open(trees["pool_l"][0] / "summary_accuracy.csv").read().splitlines()) == n + 1
    assert os.path.isfile(trees["pool_l"][0] / "overall_accuracy.txt")


def test_device_order_check_against_the_kfd_topology(tmp_path):
    """launch.check_device_order on a fake sysfs: the PCI address torch reports for HIP device i against the i-th visible GPU
    of the KFD topology that place_rank assumed"""
    from glomeruli_segmentation_amd import launch
    sysfs = tmp_path / "sys"
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    # node 0 = CPU; GPUs at 0000:05:00.0, 0000:15:00.0, 0000:85:00.0
    for i, (simd, loc) in enumerate([(0, 0), (1024, 0x0500), (1024, 0x1500), (1024, 0x8500)]):
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("simd_count %d\nlocation_id %d\ndomain 0\n" % (simd, loc))
    assert launch.gpu_pci_addresses(str(sysfs)) == ["0000:05:00.0", "0000:15:00.0", "0000:85:00.0"]
    ok, msg = launch.check_device_order(1, 0, 0x15, 0, str(sysfs), {})
    assert ok is True and "0000:15:00" in msg
    ok, msg = launch.check_device_order(1, 0, 0x85, 0, str(sysfs), {})          # HIP enumerated another GPU second
    assert ok is False and "0000:85:00" in msg and "0000:15:00.0" in msg
    ok, _ = launch.check_device_order(0, 0, 0x85, 0, str(sysfs), {"HIP_VISIBLE_DEVICES": "2,0"})
    assert ok is True                                                             # a visible-device list re-maps the ranks
    ok, _ = launch.check_device_order(5, 0, 0x05, 0, str(sysfs), {})
    assert ok is None
    ok, _ = launch.check_device_order(0, 0, 0x05, 0, str(tmp_path / "nothing"), {})
    assert ok is None


def _plan(lib, heights, widths, batch):
    import ctypes
    n = len(heights)
    hs = (ctypes.c_int * n)(*heights)
    ws = (ctypes.c_int * n)(*widths)
    nb = ctypes.c_int(0)
    assert lib.gs_plan_crop_batches(hs, ws, n, batch, None, 0, ctypes.byref(nb)) == 0
    starts = (ctypes.c_int * (nb.value + 1))()
    assert lib.gs_plan_crop_batches(hs, ws, n, batch, starts, nb.value + 1, ctypes.byref(nb)) == 0
    return list(starts)


def test_crop_batch_plan_never_exceeds_the_callers_batch():
    """the split arithmetic of gs_espnet_segment_crops_host for EVERY (batch, list length) up to four batches and beyond:
    round 4's short-list split let lists of 3.5 to 4 batches grow past the caller's batch -- and past the 64-entry descriptor
    table (a host stack overflow at e.g. batch 64, 230 crops)"""
    from glomeruli_segmentation_amd import _lib
    lib = _lib.load()
    for batch in list(range(1, 65)) + [65, 100, 1000]:
        cap = min(batch, _lib.MAX_CROPS_PER_CALL)
        for n in list(range(1, 4 * cap + 10)) + [1000]:
            starts = _plan(lib, [40] * n, [50] * n, batch)
            sizes = np.diff(starts)
            assert starts[0] == 0 and starts[-1] == n and (sizes > 0).all(), (batch, n, starts)
            assert sizes.max() <= min(cap, n), (batch, n, sizes)
            if n >= 4 * cap:                              # long lists: full batches
                assert (sizes[:-1] == cap).all(), (batch, n, sizes)
            elif n >= 32 or n > cap:                      # short lists are cut into (at least) four
                assert len(sizes) >= min(4, -(-n // 8)), (batch, n, sizes)
    # the cases the review named
    assert max(np.diff(_plan(lib, [30] * 230, [30] * 230, 64))) <= 64
    assert max(np.diff(_plan(lib, [30] * 120, [30] * 120, 32))) <= 32
    assert max(np.diff(_plan(lib, [30] * 227, [30] * 227, 57))) <= 57
    # the shape round 4 measured is kept where it fits: 56 crops at batch 32 -> 8 + 16 + 16 + 16
    assert list(np.diff(_plan(lib, [30] * 56, [30] * 56, 32))) == [8, 16, 16, 16]
    # byte cap: 4000 x 7000 crops (84 MB of pixels each) go three to a batch, one oversize crop alone
    assert list(np.diff(_plan(lib, [4000] * 7, [7000] * 7, 32))) == [3, 3, 1]
    assert list(np.diff(_plan(lib, [12000, 10, 10], [12000, 10, 10], 32))) == [1, 2]
    import ctypes
    nb = ctypes.c_int(0)
    assert lib.gs_plan_crop_batches((ctypes.c_int * 1)(0), (ctypes.c_int * 1)(5), 1, 4, None, 0, ctypes.byref(nb)) != 0   # bad size
    assert lib.gs_host_block_is_pinned(None, 10) == 0


# ------------------------------------------------------------------------------------------------------------------
# Eight-rank rehearsals on the CPU (gloo): the rank-count-dependent control flow of everything the driver will start with
# `--gpus 8` -- spawn, rendezvous, uneven and empty ranges, gathers / reductions, one rank failing, CPU placement -- so that the
# first real eight-GPU run cannot die on arithmetic that only shows with more than two ranks.  No scaling is simulated.
def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(OMP_NUM_THREADS="1", **extra)
    return env


def _last_json(text):
    import json
    return json.loads([l for l in text.strip().splitlines() if l.startswith("{")][-1])


def test_eight_ranks_bench_dry_run_and_one_rank_failing():
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _last_json(p.stdout)
    assert j["n_gpus"] == 8 and j["scaling"] == "weak" and len(j["per_rank_ms_per_step"]) == 8
    assert j["config"]["parallelism"] == "tile-range per rank x8" and j["config"]["global_batch"] == 8 * 32
    assert j["pixel_totals_all_ranks"] == [8 * 3] * 5              # the dry engine counts one per class per step: all 8 ranks reduced
    assert len(j["host_pipeline"]["per_rank_patches_per_s"]) == 8
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run",
                        "--fail-rank", "5"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 3 and time.time() - t0 < 120 and "rank 5 of 8 exited with code 3" in p.stderr, (p.returncode, p.stderr[-1500:])


def test_eight_ranks_slide_bench_equals_one_rank():
    """cfg 4's tool with CPU stand-ins: 36 windows and 56 crops over 8 ranks (4 / 5 windows, 7 crops per rank) give the slide map
    and the class totals of one rank; with a slide of four windows some ranks' ranges are EMPTY"""
    tool = os.path.join(REPO, "tools", "bench_slide.py")
    out = {}
    for size, gpus in ((40000, 1), (40000, 8), (12000, 1), (12000, 8)):
        p = subprocess.run([sys.executable, tool, "--dry-run", "--size", str(size), "--gpus", str(gpus)], env=_clean_env(),
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        out[size, gpus] = _last_json(p.stdout)
    j1, j8 = out[40000, 1], out[40000, 8]
    assert j8["windows"] == 36 and j8["crops"] == 56
    assert [b - a for a, b in j8["window_ranges"]] == [4, 5, 4, 5, 4, 5, 4, 5] and j8["window_ranges"][-1][1] == 36
    assert [b - a for a, b in j8["crop_ranges"]] == [7] * 8
    assert j8["pixel_totals"] == j1["pixel_totals"] and j8["map_nonzero"] == j1["map_nonzero"] and sum(j1["pixel_totals"]) > 0
    s1, s8 = out[12000, 1], out[12000, 8]
    assert s8["windows"] < 8 and any(a == b for a, b in s8["window_ranges"])                 # ranks without a window
    assert sum(b - a for a, b in s8["window_ranges"]) == s8["windows"] and sum(b - a for a, b in s8["crop_ranges"]) == s8["crops"]
    assert s8["pixel_totals"] == s1["pixel_totals"] and s8["map_nonzero"] == s1["map_nonzero"]


def test_eight_ranks_ensemble_bench_equals_one_rank():
    """cfg 5's tool with CPU stand-ins: 8 slides over 8 ranks (one each) and 3 slides over 8 ranks (five ranks with nothing to
    do) give one rank's per-slide totals"""
    tool = os.path.join(REPO, "tools", "bench_ensemble.py")
    out = {}
    for slides, gpus in ((8, 8), (3, 8), (8, 1)):
        p = subprocess.run([sys.executable, tool, "--dry-run", "--size", "12000", "--slides", str(slides), "--gpus", str(gpus)],
                           env=_clean_env(), capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        out[slides, gpus] = _last_json(p.stdout)
    assert out[8, 8]["pixel_totals_per_slide"] == out[8, 1]["pixel_totals_per_slide"] and len(out[8, 8]["pixel_totals_per_slide"]) == 8
    assert out[3, 8]["pixel_totals_per_slide"] == out[8, 1]["pixel_totals_per_slide"][:3]
    assert all(sum(row) > 0 for row in out[8, 8]["pixel_totals_per_slide"])


def _spawn_env():
    saved = {k: os.environ.get(k) for k in ("GLOMSEG_DIST_BACKEND", "GS_TEST_FAIL_RANK", "MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK",
                                            "OMP_NUM_THREADS")}
    for k in ("MASTER_PORT", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    os.environ["GLOMSEG_DIST_BACKEND"] = "gloo"
    os.environ["OMP_NUM_THREADS"] = "1"
    return saved


def _restore_env(saved):
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_eight_ranks_segment_cli_writes_one_set_of_files(tmp_path):
    """the segment command line's sharding, gathers and file writing as 8 gloo ranks with a CPU stand-in for the engine: 56 crops
    (7 per rank) and 5 crops (three ranks with an empty range) write byte for byte the files of one process; one rank failing
    alone ends the job"""
    import filecmp
    import io
    from PIL import Image
    from glomeruli_segmentation_amd import launch
    from glomeruli_segmentation_amd.synth import synth_tile
    helper = os.path.join(REPO, "tests", "helpers", "segment_stub_rank.py")
    saved = _spawn_env()
    try:
        for n in (56, 5):
            rgb = tmp_path / ("rgb%d" % n)
            for k in range(n):
                d = rgb / ("P%d" % (k % 3))
                d.mkdir(parents=True, exist_ok=True)
                Image.fromarray(np.ascontiguousarray(synth_tile(k, 24 + k % 5, 40 + k % 7, blobs=2)[:, :, ::-1])).save(d / ("xmin%d_ymin0_xmax9_ymax9.PNG" % k))
            argv = ["--rgb_data_dir", str(rgb), "--weights", "unused", "--mean", "1", "2", "3", "--std", "1", "2", "3", "--inWidth", "64",
                    "--inHeight", "32", "--batch", "4", "--colored", "--overlay", "--cityFormat", "--workers", "1"]
            os.environ["GS_TEST_FAIL_RANK"] = "none"
            os.environ.pop("MASTER_PORT", None)
            out8, err8 = io.StringIO(), io.StringIO()
            assert launch.spawn_ranks(helper, argv + ["--savedir", str(tmp_path / ("out8_%d" % n))], 8, out=out8, err=err8) == 0, err8.getvalue()[-2000:]
            p = subprocess.run([sys.executable, helper] + argv + ["--savedir", str(tmp_path / ("out1_%d" % n))], env=_clean_env(),
                               capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            a, b = tmp_path / ("out1_%d" % n), tmp_path / ("out8_%d" % n)
            fa = sorted(os.path.relpath(os.path.join(d, f), a) for d, _, fs in os.walk(a) for f in fs)
            fb = sorted(os.path.relpath(os.path.join(d, f), b) for d, _, fs in os.walk(b) for f in fs)
            assert fa == fb and len(fa) >= 4 * n + 1
            assert all(filecmp.cmp(a / f, b / f, shallow=False) for f in fa)
            assert len(open(b / "summary_pixel.csv").read().splitlines()) == n + 1
        os.environ["GS_TEST_FAIL_RANK"] = "6"
        os.environ.pop("MASTER_PORT", None)
        err = io.StringIO()
        rc = launch.spawn_ranks(helper, argv + ["--savedir", str(tmp_path / "out_fail")], 8, out=io.StringIO(), err=err)
        assert rc == 1 and "rank 6 of 8 exited with code 1" in err.getvalue() and "fails on purpose" in err.getvalue()
    finally:
        _restore_env(saved)


def test_eight_ranks_detect_cli_equals_one_process(tmp_path):
    """the detect command line with the stub detector as 8 gloo ranks: 21 windows (2 / 3 per rank) write the CSV of one process,
    row for row; rank 3 failing alone ends the job at once"""
    import io
    import time
    from glomeruli_segmentation_amd import launch
    data_dir, tl, _ = _png_slide_tree(tmp_path)
    helper = os.path.join(REPO, "tests", "helpers", "detect_stub_rank.py")
    argv = ["--target_list", tl, "--data_dir", data_dir, "--staining", "OPT_PAS", "--window_size", "2000", "--overlap_ratio", "0.1", "--batch", "2"]
    saved = _spawn_env()
    try:
        os.environ["GS_TEST_FAIL_RANK"] = "none"
        err = io.StringIO()
        assert launch.spawn_ranks(helper, argv + ["--output_dir", str(tmp_path / "out8")], 8, out=io.StringIO(), err=err) == 0, err.getvalue()[-2000:]
        p = subprocess.run([sys.executable, helper] + argv + ["--output_dir", str(tmp_path / "out1")], env=_clean_env(GS_TEST_FAIL_RANK="none"),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]

        def rows(d):      # (field 4 is the wall-clock time of the row)
            return [r.split(",")[:4] + r.split(",")[5:] for r in open(d / "OPT_PAS_GlomusList.csv").read().splitlines()]
        assert rows(tmp_path / "out8") == rows(tmp_path / "out1") and len(rows(tmp_path / "out8")) == 21
        os.environ["GS_TEST_FAIL_RANK"] = "3"
        os.environ.pop("MASTER_PORT", None)
        t0 = time.time()
        err = io.StringIO()
        rc = launch.spawn_ranks(helper, argv + ["--output_dir", str(tmp_path / "outf")], 8, out=io.StringIO(), err=err)
        assert rc == 1 and time.time() - t0 < 120 and "rank 3 of 8 exited with code 1" in err.getvalue()
    finally:
        _restore_env(saved)


def test_eight_ranks_split_sixteen_cpus(tmp_path):
    """launch.rank_cpus with eight ranks: 16 allowed CPUs and no topology -> two each, disjoint, all used; eight GPUs on two
    NUMA nodes of a fake sysfs -> every rank gets a quarter of its node's allowed CPUs; fewer CPUs than ranks still gives every
    rank something to run on"""
    from glomeruli_segmentation_amd import launch
    got = [launch.rank_cpus(r, 8, set(range(16)), str(tmp_path / "none"), {}) for r in range(8)]
    assert got == [[2 * r, 2 * r + 1] for r in range(8)]
    sysfs = tmp_path / "sys"
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for k in range(8):
        bus, numa = 0x05 + 0x10 * k, k // 4
        d = nodes / str(k + 1)
        d.mkdir()
        d.joinpath("properties").write_text("simd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        pci = sysfs / "bus" / "pci" / "devices" / ("0000:%02x:00.0" % bus)
        pci.mkdir(parents=True)
        pci.joinpath("numa_node").write_text("%d\n" % numa)
    for n, cl in ((0, "0-31"), (1, "32-63")):
        d = sysfs / "devices" / "system" / "node" / ("node%d" % n)
        d.mkdir(parents=True)
        d.joinpath("cpulist").write_text(cl + "\n")
    got = [launch.rank_cpus(r, 8, set(range(64)), str(sysfs), {}) for r in range(8)]
    assert [len(g) for g in got] == [8] * 8 and sorted(c for g in got for c in g) == list(range(64))
    assert all(max(g) < 32 for g in got[:4]) and all(min(g) >= 32 for g in got[4:])
    # a cgroup of 16 CPUs, all on node 0: ranks of node 0 share them, ranks of node 1 (no allowed CPU there) still get some
    got = [launch.rank_cpus(r, 8, set(range(16)), str(sysfs), {}) for r in range(8)]
    assert all(len(g) >= 1 for g in got) and all(set(g) <= set(range(16)) for g in got)
    got = [launch.rank_cpus(r, 8, {0, 1, 2}, str(tmp_path / "none"), {}) for r in range(8)]
    assert all(len(g) >= 1 and set(g) <= {0, 1, 2} for g in got)
