"""One process per GPU on one node: the parent side (spawn, watch, fail fast) and the child side (CPU placement)
shared by bench.py, tools/bench_slide.py, tools/bench_ensemble.py and the CLIs.

Nothing in this module imports torch or touches the GPU: the parent must stay GPU-free (its children are the first
processes of the job to initialise the device, and a process that has initialised the GPU is never re-executed), and the
child sets its CPU affinity BEFORE its first GPU call so that pinned staging buffers land on its GPU's NUMA node.

The reference has no distributed code; the partitioning is SURVEY 8e.
"""
import os
import socket
import subprocess
import sys
import tempfile
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(script, argv, n, poll_s=0.05, grace_s=5.0, out=None, err=None):
    """Start `python script argv...` (or, when `script` is a list, that command prefix -- e.g. [sys.executable, "-m",
    "glomeruli_segmentation_amd.segment"]) n times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's stdout, and
    watch ALL children: as soon as one exits non-zero the others are terminated (they would sit in a rendezvous or a
    collective until its timeout, holding their GPUs) and the failing rank's stderr tail is reported.  Returns the exit code
    for the parent (0 only if every rank returned 0).  Fresh child processes only; nothing is restarted."""
    out = out or sys.stdout
    err = err or sys.stderr
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = str(n)
    env["LOCAL_WORLD_SIZE"] = str(n)
    procs, logs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        log = tempfile.TemporaryFile()
        logs.append(log)
        cmd = list(script) if isinstance(script, (list, tuple)) else [sys.executable, os.path.abspath(script)]
        procs.append(subprocess.Popen(cmd + list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=log))

    def tail(f, limit=3000):
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(max(0, size - limit))
        return f.read().decode("utf-8", "replace")

    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        time.sleep(poll_s)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    out.write(out0.read().decode("utf-8", "replace"))
    out.flush()
    if failed is not None:
        r, rc = failed
        print("rank %d of %d exited with code %s; the other ranks were terminated.  Its stderr ends:\n%s" % (r, n, rc, tail(logs[r])),
              file=err)
        return rc if isinstance(rc, int) and 0 < rc < 256 else 1
    # successful runs still pass the ranks' warnings on
    for r, log in enumerate(logs):
        t = tail(log, 1500)
        if t.strip() and r == 0:
            err.write(t)
    return 0


# --------------------------------------------------------------------------------------------- CPU placement
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _parse_cpulist(text):
    cpus = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_pci_addresses(sysfs="/sys"):
    """PCI address "dddd:bb:dd.f" of every GPU in KFD topology order, from sysfs alone -- no GPU call.  [] when the topology
    cannot be read."""
    root = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted((int(d) for d in os.listdir(root) if d.isdigit()))
    except OSError:
        return []
    out = []
    for i in ids:
        props = {}
        for line in (_read(os.path.join(root, str(i), "properties")) or "").splitlines():
            kv = line.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        try:
            if int(props.get("simd_count", "0")) == 0:
                continue          # a CPU node
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        except ValueError:
            continue
        out.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 0x7))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in KFD topology order (= HIP device order when HIP_VISIBLE_DEVICES does not permute it; checked
    after the device is initialised by check_device_order).  [] when the topology cannot be read; -1 for a GPU whose node is
    unknown."""
    nodes = []
    for bdf in gpu_pci_addresses(sysfs):
        numa = _read(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node"))
        try:
            nodes.append(int(numa))
        except (TypeError, ValueError):
            nodes.append(-1)
    return nodes


def check_device_order(local_rank, pci_domain, pci_bus, pci_device, sysfs="/sys", env=None):
    """place_rank chose this rank's CPUs BEFORE the first GPU call, from the assumption that HIP device `local_rank` is the
    local_rank-th visible GPU of the KFD topology.  Once the device is initialised its PCI address is known
    (torch.cuda.get_device_properties(i).pci_domain_id / pci_bus_id / pci_device_id): compare.  Returns (ok, message); ok is
    None when the topology cannot be read.  A mismatch costs speed (staging buffers on the far NUMA node), not correctness, so
    callers log it and go on."""
    env = os.environ if env is None else env
    phys = gpu_pci_addresses(sysfs)
    if not phys:
        return None, "KFD topology unreadable: device order not checked"
    vis = visible_gpu_indices(len(phys), env)
    if local_rank >= len(vis):
        return None, "local rank %d beyond the %d visible GPUs of the topology" % (local_rank, len(vis))
    want = phys[vis[local_rank]]
    have = "%04x:%02x:%02x" % (pci_domain, pci_bus, pci_device)
    if want.rsplit(".", 1)[0] == have:
        return True, "HIP device %d is %s, as the KFD topology order says" % (local_rank, have)
    return False, ("HIP device %d is at PCI %s but the KFD topology order puts %s there: this rank's CPU placement (NUMA node) was "
                   "chosen for the wrong GPU" % (local_rank, have, want))


def visible_gpu_indices(n_physical, env=None):
    """physical (KFD-order) index of every visible device, from ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when they are plain index lists (the first applies to the KFD order, the other two to what it leaves
    visible); the identity when unset or not numeric (UUIDs)."""
    env = os.environ if env is None else env
    vis = list(range(n_physical))
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(name)
        if not v:
            continue
        try:
            idx = [int(t) for t in v.split(",") if t.strip() != ""]
        except ValueError:
            continue
        if all(0 <= i < len(vis) for i in idx):
            vis = [vis[i] for i in idx]
        if name != "ROCR_VISIBLE_DEVICES":
            break          # HIP_ and CUDA_VISIBLE_DEVICES are aliases: the first one set wins
    return vis


def rank_cpus(local_rank, local_world, allowed=None, sysfs="/sys", env=None):
    """The CPUs rank `local_rank` of `local_world` should run on: the allowed CPUs of its GPU's NUMA node, shared evenly with
    the other ranks whose GPUs sit on the same node; an even split of the allowed set when the topology says nothing."""
    if allowed is None:
        try:
            allowed = os.sched_getaffinity(0)
        except AttributeError:
            return None
    allowed = sorted(allowed)
    if local_world <= 1 or not allowed:
        return allowed
    phys = gpu_numa_nodes(sysfs)
    numa = [phys[i] for i in visible_gpu_indices(len(phys), env)]      # NUMA node of visible device 0, 1, ...
    if len(numa) >= local_world and numa[local_rank] >= 0:
        node = numa[local_rank]
        cpus = sorted(_parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist"))) & set(allowed))
        peers = [r for r in range(local_world) if numa[r] == node]
        if len(cpus) >= len(peers):
            k = peers.index(local_rank)
            lo, hi = len(cpus) * k // len(peers), len(cpus) * (k + 1) // len(peers)
            return cpus[lo:hi]
    lo, hi = len(allowed) * local_rank // local_world, len(allowed) * (local_rank + 1) // local_world
    return allowed[lo:hi] or allowed


def place_rank(local_rank=None, local_world=None):
    """Child side, before the first GPU call: pin this process to its GPU's share of the host cores.  Returns the CPU list
    (None when the platform has no affinity call).  A job with one rank is left alone."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else local_rank
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else local_world
    if local_world <= 1:
        return None
    cpus = rank_cpus(local_rank, local_world)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except (AttributeError, OSError):
            return None
    return cpus
