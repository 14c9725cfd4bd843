#!/usr/bin/env python3
"""Headline benchmark: ESPNet (p=2, q=8, 5 classes) patches/sec on 1024x512 BGR tiles.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch of 32 synthetic uint8 tiles already resident in
HBM: normalise -> ESPNet forward -> argmax -> uint8 masks + per-class pixel counts
(reference loop body: module/espnet/test/VisualizeResults_iou.py:107-128,151-155).  Steps rotate
through four distinct batches, so the inputs of a step are not the ones the caches saw last.

N > 1 is one process per GPU over torch.distributed (RCCL): every rank owns its own tile range (weak
scaling) and the only exchange is one all-reduce of the per-class pixel totals at the end of the
timed region.  Under `python -m torch.distributed.run` the ranks come from the launcher's environment;
started plainly with --gpus N > 1 this process -- before it makes any GPU call -- starts N children
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, waits for them and relays rank 0's JSON line.

Two batches are in flight at a time: consecutive steps go to the engine's two LANES (two activation workspaces, each on
its own HIP stream; include/glomseg.h).  The kernels of one batch run strictly one after the other, so what overlaps is
the tail of one batch's kernel with the head of another's.  `value` is that declared configuration (`batches_in_flight` = 2:
`two_lanes`); the same loop timed with one batch in flight is reported beside it (`single_lane`) and is never `value`: two
lanes win by 2-4 % on most boxes of the pool, on some the single stream does by ~1 %.

The timed K-step loop is repeated (default 5 times, barrier + synchronize on both sides of each) and
`value` is the median repeat; every repeat's time is in the line.  `value` is the HBM-resident rate;
`host_pipeline` (pinned host tiles in -> pinned host masks out, every rank with its own staging
buffers) is measured for every N beside it.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BATCH = 32
NBATCH = 4                       # distinct batches the steps rotate through
LANES = 2                        # batches in flight (engine lanes)
H, W = 512, 1024
FLOP_PER_TILE = 7.267e9          # conv/deconv MACs x 2, unpadded channel counts
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
DOMINANT = "conv_l3_esp_branches"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="times the K-step loop is timed (median reported)")
    ap.add_argument("--lanes", type=int, default=LANES, help="batches in flight (1 = one workspace, one stream: what the "
                                                             "rocprofv3 passes of tools/profile_round.sh use, so that "
                                                             "kernel durations are not stretched by a co-running batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-pipeline", action="store_true")
    ap.add_argument("--host-batches", type=int, default=224,
                    help="batches of the pinned-host pipeline leg (SURVEY 8d: >= 200 after >= 20 warm-up), run as calls of 32 "
                         "batches cycling one set of pinned buffers")
    ap.add_argument("--host-min-seconds", type=float, default=6.5,
                    help="the pinned-host leg keeps going for at least this long (continuous GPU work a periodic sampler can see)")
    ap.add_argument("--no-real-crops", action="store_true")
    ap.add_argument("--fail-rank", type=int, default=-1, help="test hook (--dry-run only): this rank exits with code 3 after "
                                                              "the rendezvous, to exercise the parent's fail-fast path")
    ap.add_argument("--dry-run", action="store_true",
                    help="control flow only (spawn, rendezvous, reductions, JSON) with a no-op step on CPU/gloo: "
                         "what the CPU test suite runs; never a measurement")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """Parent of a plain `bench.py --gpus N`: start one child per GPU and watch all of them (a rank that dies takes the job
    down at once, with its stderr tail, instead of leaving the others in a collective until a timeout).  Nothing here
    touches the GPU (no torch import), so the children are the first processes of this job to initialise it."""
    from glomeruli_segmentation_amd.launch import spawn_ranks as spawn
    return spawn(os.path.abspath(__file__), sys.argv[1:], args.gpus)


# ---------------------------------------------------------------------------------------------
def load_weights():
    import numpy as np
    z = np.load(os.path.join(REPO, "tests", "golden", "weights_fold1.npz"))
    return {k: z[k] for k in z.files}


def make_batches(rank):
    """NBATCH x BATCH tiles; the first four tiles of rank 0 are the golden seeds 0..3 (in-line parity)."""
    import numpy as np
    from glomeruli_segmentation_amd.synth import synth_tile
    base = rank * NBATCH * BATCH
    return np.stack([synth_tile(base + i) for i in range(NBATCH * BATCH)]).reshape(NBATCH, BATCH, H, W, 3)


def host_cores():
    """CPU threads this process may really use: the cgroup quota when there is one (a GPU box hands
    each job a share of the host), else the affinity mask."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def _cgroup_cpu():
    """what the kernel says about this process's CPU share: the cgroup quota and its throttling counters"""
    out = {}
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        out["cpu_max"] = "%s %s" % (quota, period)
    except (OSError, ValueError):
        out["cpu_max"] = None
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            st = dict(line.split() for line in f if len(line.split()) == 2)
        for k in ("nr_periods", "nr_throttled", "throttled_usec"):
            if k in st:
                out[k] = int(st[k])
    except OSError:
        pass
    return out


def cpu_baseline(sd, tiles, mean, std):
    """The torch-operator port of the reference graph on this box's host cores; bounded sample.

    `value` is bound to BATCH 1 -- the batch size of the reference's own loop (VisualizeResults_iou.py:119-123: one crop per
    `model(img_variable)`) -- with batch 4 beside it, each as 3 warm-up + 10 timed forwards.  The line also
    carries what is needed to read the two figures against each other: torch's thread count, the affinity mask, the cgroup CPU
    quota, every iteration's time and the cgroup's throttling counters over each leg (a process whose OpenMP team spins on
    more runnable threads than its quota pays for them in throttled periods; the batch-4 forward holds its team ~4x longer
    per call)."""
    import torch
    from oracle import espnet_torch_port as port   # bench's cpu_baseline leg only
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    cores = host_cores()
    torch.set_num_threads(cores)
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1

    def run(bs, warm=3, timed=10):
        x = port.preprocess(tiles[:bs], mean, std)
        for _ in range(warm):
            port.espnet_forward(x, tsd)
        c0 = _cgroup_cpu()
        its = []
        for _ in range(timed):
            t0 = time.perf_counter()
            port.espnet_forward(x, tsd)
            its.append(time.perf_counter() - t0)
        c1 = _cgroup_cpu()
        leg = {"value": round(bs * timed / sum(its), 3), "unit": "patches/s", "batch": bs,
               "iteration_ms": [round(1e3 * t, 1) for t in its],
               "sample": "%d warm-up + %d timed batch-%d forwards of the 1024x512 workload through torch CPU ops "
                         "(oracle/espnet_torch_port.py)" % (warm, timed, bs)}
        for k in ("nr_periods", "nr_throttled", "throttled_usec"):
            if k in c0 and k in c1:
                leg["cgroup_" + k] = c1[k] - c0[k]
        return leg

    b1 = run(1)     # what the reference's loop runs
    b4 = run(4)
    out = {"value": b1["value"], "unit": "patches/s", "cores": cores, "kind": "port", "batch": 1, "sample": b1["sample"],
           "iteration_ms": b1["iteration_ms"], "torch_num_threads": torch.get_num_threads(), "affinity_cpus": affinity,
           "cgroup_cpu_max": _cgroup_cpu().get("cpu_max"), "batch4": b4,
           "note": "value = batch 1, the batch size of the reference's own loop; batch 4 beside it (BASELINE.md 4)"}
    for k in ("cgroup_nr_periods", "cgroup_nr_throttled", "cgroup_throttled_usec"):
        if k in b1:
            out[k] = b1[k]
    return out


def parity_vs_golden(mask_np):
    import numpy as np
    from oracle import espnet_oracle as orc       # checker only
    z = np.load(os.path.join(REPO, "tests", "golden", "masks_fold1.npz"))
    conf = np.zeros((5, 5), dtype=np.int64)
    for s in range(4):
        conf += orc.confusion(mask_np[s], z["mask_%d" % s])
    return {"miou_vs_reference": round(orc.present_class_miou(conf), 6),
            "pixel_agreement": round(float(np.trace(conf)) / float(conf.sum()), 7), "tiles": 4}


def traffic_of_dominant():
    import glob
    cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")))   # newest round's rocprofv3 --pmc passes
    if not cands:
        return None, None
    with open(cands[-1]) as f:
        tj = json.load(f)
    if tj.get("kernel") != DOMINANT:
        return None, None
    return tj["traffic_bytes_per_launch"], {"algorithmic_bytes_per_launch": tj["algorithmic_bytes_per_launch"],
                                            "source": tj["source"], "correction": tj.get("correction"),
                                            "file": os.path.basename(cands[-1])}


class DryEngine:
    """--dry-run stand-in: the same calls, no device work (control-flow tests on CPU)."""

    def __init__(self):
        self.on = False

    def reserve(self, *a):
        pass

    def segment(self, tiles, mean, std, out_mask=None, out_hist=None, lane=None):
        out_hist.fill_(1)

    def wait_lanes(self):
        pass

    def profile(self, on):
        self.on = on

    def profile_read(self):
        return [{"name": DOMINANT, "total_ms": 1.0, "launches": 1, "flops_per_tile": 1.0}]

    def segment_host(self, tiles, mean, std, batch=32, out_masks=None, out_hist=None):
        return out_masks.numpy(), out_hist.numpy()


def run_rank(args):
    import numpy as np
    import torch
    global LANES
    LANES = max(1, min(int(args.lanes), 4))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from glomeruli_segmentation_amd.launch import place_rank
    cpus = place_rank()        # before the first GPU call: this rank's share of its GPU's NUMA node (no-op for one rank)
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    # rehearsal knobs for a one-GPU box (never set by the driver): GS_BENCH_BACKEND=gloo GS_BENCH_ONE_GPU=1 run
    # the N-rank control flow with every rank on device 0 and the tiny reductions staged through host memory
    backend = "gloo" if args.dry_run else os.environ.get("GS_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("GS_BENCH_ONE_GPU") == "1":
            local = 0
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
    elif not args.dry_run:
        torch.cuda.set_device(0)
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", torch.cuda.current_device())
    if args.dry_run and args.fail_rank == rank:
        print("bench.py: rank %d fails on purpose (--fail-rank)" % rank, file=sys.stderr)
        os._exit(3)

    def sync():
        if not args.dry_run:
            torch.cuda.synchronize()

    from glomeruli_segmentation_amd.shard import all_reduce_any, log_device_order
    if world > 1 and os.environ.get("GS_BENCH_ONE_GPU") != "1" and not getattr(args, "dry_run", False):
        log_device_order(local)      # is HIP device `local` the GPU place_rank pinned this rank's CPUs for? (stderr, never fatal)

    def all_reduce(t, op=None):      # one helper owns the backend choice (device tensors as they are under RCCL, via the host under gloo)
        all_reduce_any(t, dist, op)

    def gather_f64(x):
        """one float64 per rank -> list on every rank"""
        if dist is None:
            return [float(x)]
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = x
        all_reduce(t)
        return [float(v) for v in t.tolist()]

    from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD
    mean, std = FOLD_MEAN_STD[1]
    if args.dry_run:
        sd = None
        eng = DryEngine()
        tiles_np = np.zeros((NBATCH, 1, 8, 8, 3), dtype=np.uint8)
    else:
        from glomeruli_segmentation_amd.engine import EspnetEngine
        sd = load_weights()
        eng = EspnetEngine(sd, classes=5, p=2, q=8, lanes=LANES)
        eng.reserve(BATCH, H, W)
        tiles_np = make_batches(rank)
    tiles = torch.from_numpy(tiles_np).to(dev)                       # resident before any timed region
    mask = torch.zeros((NBATCH,) + tiles_np.shape[1:4], dtype=torch.uint8, device=dev)
    # per-class counts of every step of a timed loop (each step has its own [batch, 5] slice); summed once at its end
    nslots = max(args.steps, args.warmup, NBATCH)
    hist = torch.zeros((nslots, tiles_np.shape[1], 5), dtype=torch.int64, device=dev)
    totals = torch.zeros(5, dtype=torch.int64, device=dev)
    counter = [0]
    mode = {"lanes": LANES}

    def step():
        i = counter[0]
        counter[0] += 1
        b = i % NBATCH
        h = hist[i % nslots]
        if mode["lanes"] == 1:
            eng.segment(tiles[b], mean, std, out_mask=mask[b], out_hist=h)
            return
        # consecutive steps alternate between the lanes: two batches in flight
        eng.segment(tiles[b], mean, std, out_mask=mask[b], out_hist=h, lane=i % LANES)

    for _ in range(max(args.warmup, NBATCH)):   # every batch once; also loads torch's own reduce/add code objects
        step()
    sync()

    def timed(steps):
        counter[0] = 0
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        eng.wait_lanes()
        torch.sum(hist[:steps], (0, 1), out=totals)   # slide-level per-class pixel totals of this rank
        if dist is not None:
            all_reduce(totals)             # the one exchange: slide-level per-class pixel totals
        sync()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0

    # the reported numbers: no instrumentation in the stream; max over ranks per repeat, median over repeats
    def reduce_max(local):
        out = []
        for el in local:
            if dist is not None:
                tmax = torch.tensor([el], dtype=torch.float64, device=dev)
                all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            out.append(el)
        return out

    reps_local = [timed(args.steps) for _ in range(max(args.repeats, 1))]
    mode["lanes"] = 1                      # the same loop with one batch in flight
    timed(args.steps)
    single_local = [timed(args.steps) for _ in range(max(args.repeats, 1))]
    mode["lanes"] = LANES
    timed(args.steps)                      # leaves `totals` from a pass in the default mode
    reps_lanes, reps_single = reduce_max(reps_local), reduce_max(single_local)
    elapsed_lanes, single = float(np.median(reps_lanes)), float(np.median(reps_single))
    # `value` is bound to ONE declared configuration -- LANES batches in flight (two by default) -- not to the better of two noisy
    # measurements; the single-stream figure is reported beside it
    reps, elapsed = reps_lanes, elapsed_lanes
    in_flight = LANES
    per_rank_ms = gather_f64(float(np.median(reps_local)) / args.steps * 1e3)
    pixel_totals = [int(v) for v in totals.tolist()]   # all-reduced in the last repeat

    # same K steps again, one batch in flight, with a HIP event pair around every kernel on the launch stream: per-kernel
    # durations for the roofline line (the event packets lengthen kernel boundaries, so this pass is
    # not the one quoted as throughput; its wall time is reported beside it)
    mode["lanes"] = 1
    eng.profile(True)
    elapsed_prof = timed(args.steps)
    prof = eng.profile_read()
    eng.profile(False)
    mode["lanes"] = LANES

    # PCIe-inclusive rate (never `value`): pinned host uint8 tiles -> pinned host masks through the
    # H2D / compute / D2H pipeline of gs_espnet_segment_host, every rank with its own buffers
    host = None
    if not args.no_host_pipeline:
        nb = tiles_np.shape[1]
        per_call = 32 if not args.dry_run else NBATCH                 # batches per call: one set of pinned buffers, cycled
        calls = max(1, -(-args.host_batches // per_call)) if not args.dry_run else 1
        flat = tiles_np.reshape((-1,) + tiles_np.shape[2:])
        host_tiles = torch.from_numpy(np.concatenate([flat] * (per_call // NBATCH)))
        om = torch.zeros(host_tiles.shape[:3], dtype=torch.uint8)
        oh = torch.zeros((host_tiles.shape[0], 5), dtype=torch.int64)
        if not args.dry_run:
            host_tiles, om, oh = host_tiles.pin_memory(), om.pin_memory(), oh.pin_memory()   # caller-owned pinned buffers
        eng.segment_host(host_tiles, mean, std, batch=nb, out_masks=om, out_hist=oh)          # warm-up: 32 batches
        if dist is not None:
            dist.barrier()
        # at least `calls` calls (>= 200 batches), and at least HOST_LEG_MIN_S of continuous GPU work: a sampler that looks
        # at the GPU every few seconds (the driver's smi sampler) then sees it busy, and the figure is steadier
        min_s = 0.0 if args.dry_run else args.host_min_seconds
        t0 = time.perf_counter()
        done = 0
        while done < calls or time.perf_counter() - t0 < min_s:
            hm, _ = eng.segment_host(host_tiles, mean, std, batch=nb, out_masks=om, out_hist=oh)
            done += 1
        el_h = time.perf_counter() - t0
        calls = done
        n_host = int(host_tiles.shape[0]) * calls
        same = bool((hm[:nb] == mask[0].cpu().numpy()).all())
        # the leg runs until a wall-clock minimum, so the number of calls differs from rank to rank: gather both
        t_r, n_r = gather_f64(el_h), gather_f64(float(n_host))
        same_all = min(gather_f64(1.0 if same else 0.0)) == 1.0
        host = {"value": round(sum(n_r) / max(t_r), 1), "unit": "patches/s",
                "tiles_per_rank": [int(v) for v in n_r], "batches_per_rank": [int(v) // nb for v in n_r],
                "warmup_batches": int(host_tiles.shape[0] // nb),
                "calls": calls, "per_rank_patches_per_s": [round(n / t, 1) for n, t in zip(n_r, t_r)],
                "note": "pinned host in -> pinned host out, PCIe inclusive, every rank its own staging buffers; every call "
                        "fills and drains the pipeline once; masks equal the resident path: %s" % same_all}

    # the path REAL crops take (VisualizeResults_iou.py:100-156 with crops that are not network-sized): the 28 crop sizes of the
    # example slide, pageable numpy crops in -> crop-size maps + counts out through gs_espnet_segment_crops_host
    real = None
    if not args.no_real_crops and not args.dry_run:
        from glomeruli_segmentation_amd.synth import synth_tile
        ex = np.load(os.path.join(REPO, "tests", "golden", "merge.npz"))["example_boxes"]
        base = [synth_tile(5000 + rank * 100 + k, int(b[3] - b[1]), int(b[2] - b[0]), blobs=4) for k, b in enumerate(ex)]
        crops = base * 16                                             # 448 crops = 14 batches of 32
        eng.segment_crops(crops, mean, std, H, W, BATCH)              # warm-up: 14 batches (staging buffers, pinned output block)
        el_c, el_p = [], []
        pinned = [torch.from_numpy(c).pin_memory() for c in base] * 16
        eng.segment_crops(pinned, mean, std, H, W, BATCH)
        for _ in range(5):                                            # 5 x 14 batches each way, median call
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            rr = eng.segment_crops(crops, mean, std, H, W, BATCH)
            el_c.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            eng.segment_crops(pinned, mean, std, H, W, BATCH)
            el_p.append(time.perf_counter() - t0)
        el_c, el_p = float(np.median(el_c)), float(np.median(el_p))
        ok = all(int(rr["counts"][i].sum()) == crops[i].shape[0] * crops[i].shape[1] for i in range(len(crops)))
        real = {"value": round(world * len(crops) / max(gather_f64(el_c)), 1), "unit": "crops/s", "crops_per_rank": len(crops),
                "pinned_inputs_crops_per_s": round(world * len(crops) / max(gather_f64(el_p)), 1),
                "mean_crop_px": int(np.mean([c.shape[0] * c.shape[1] for c in base])),
                "note": "crop sizes of the example slide's 28 boxes (tests/golden/merge.npz), pageable numpy crops in, crop-size "
                        "maps and counts out (pinned); resample + forward + argmax + resize back per batch of 32; median of 5 calls "
                        "of 14 batches after a warm-up call (every call fills and drains the pipeline once); "
                        "counts cover every pixel: %s" % ok}

    rc = 0
    if rank == 0:
        n = world
        total_tiles = n * tiles_np.shape[1] * args.steps
        value = total_tiles / elapsed
        dom = next((k for k in prof if k["name"] == DOMINANT), None)
        kernels = {k["name"]: {"avg_ms": round(k["total_ms"] / max(k["launches"], 1), 4), "launches": k["launches"]}
                   for k in prof}
        roof = None
        if dom:
            traffic, traffic_detail = traffic_of_dominant()
            avg_s = dom["total_ms"] / dom["launches"] * 1e-3
            achieved = dom["flops_per_tile"] * BATCH / avg_s / 1e12
            roof = {"bound": "mfma", "kernel": DOMINANT, "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                    "traffic_is": "static: read from the committed profiles/ file named in traffic_detail (rocprofv3 --pmc passes of "
                                  "the builder's run of this command), NOT measured in this run",
                    "traffic_detail": traffic_detail,
                    "flop_per_launch": dom["flops_per_tile"] * BATCH,
                    "avg_launch_ms": round(avg_s * 1e3, 4),
                    "launches_timed": dom["launches"], "instrumented_ms_per_step": round(elapsed_prof / args.steps * 1e3, 3),
                    "whole_net_frac": round(value / n * FLOP_PER_TILE / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "whole_net_frac_single_lane": round(total_tiles / single / n * FLOP_PER_TILE / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
        out = {
            "metric": "patches/sec (1024x512 RGB) ESPNet p=2 q=8 5 classes, HBM-resident, two batches in flight", "value": round(value, 2),
            "unit": "patches/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "dry-run (no device work)" if args.dry_run else "synthetic",
            "config": {"workload": "ESPNet p=2 q=8 encoder+decoder, batch=32 synthetic 1024x512 uint8 BGR tiles per GPU, "
                                   "normalise+forward+argmax+counts, inputs resident in HBM, steps rotate through %d distinct batches, "
                                   "%d batch(es) in flight (engine lanes, one HIP stream each) for `value`; the one-batch-in-flight "
                                   "figure is reported beside it" % (NBATCH, in_flight),
                       "global_batch": n * BATCH, "tile": [H, W], "weights": "espnet_fold1 (tests/golden)",
                       "parallelism": "tile-range per rank x%d" % n},
            "batches_in_flight": in_flight,
            "two_lanes": {"value": round(total_tiles / elapsed_lanes, 2), "ms_per_step": round(elapsed_lanes / args.steps * 1e3, 3),
                          "seconds": [round(r, 6) for r in reps_lanes],
                          "note": "the K steps alternating between %d lanes (workspaces), each on its own stream" % LANES},
            "single_lane": {"value": round(total_tiles / single, 2), "ms_per_step": round(single / args.steps * 1e3, 3),
                            "seconds": [round(r, 6) for r in reps_single],
                            "note": "the same K steps with one batch in flight (one workspace, one stream)"},
            "repeats": {"n": len(reps), "statistic": "median of max-over-ranks", "seconds": [round(r, 6) for r in reps]},
            "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
            "pixel_totals_all_ranks": pixel_totals,
            "roofline": roof,
            "kernels_avg_ms": kernels,
        }
        if host is not None:
            out["host_pipeline"] = host
        if real is not None:
            out["real_crops"] = real
        if cpus is not None:
            out["rank0_cpus"] = len(cpus)
        if not args.dry_run:
            out["parity"] = parity_vs_golden(mask[0, :4].cpu().numpy())
            if n == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(sd, tiles_np[0], mean, std)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
