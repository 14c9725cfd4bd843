"""Builds libglomseg.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m glomeruli_segmentation_amd.build [--force] [--out PATH] [-- extra hipcc flags]

The .so is git-ignored but travels to the GPU box with the working tree.  hipcc cross-compiles
gfx950 code objects without a GPU.  `--out` + extra flags build an experiment variant of the same
sources (e.g. `--out variants_so/libglomseg_nofuse.so -- -DCFG_FUSE_L3=0`), selected at run time
with GLOMSEG_EXPERIMENT=1 GLOMSEG_LIB=...; `-DGS_DIAG` adds the timing / stamp variants the product build leaves out.
"""
import os
import subprocess
import sys
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libglomseg.so")
SOURCES = ["espnet.hip", "dec_tail.hip", "crops.hip", "detect_ops.hip", "detector.hip", "contours.cpp"]
HEADERS = ["gs_internal.h", "conv_mfma.h", "dec_tail.h", "dec_tail_args.h", "espnet_kernels.h", "crop_sample.h", "crop_plan.h", "host_copy.h", "host_jobs.h",
           os.path.join("..", "..", "include", "glomseg.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result",
         "-Wno-unused-value"]      # extra flags only through the command line (`-- ...` with --out): no environment knob


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale(lib):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in _sources() + HEADERS] + [os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False, out=None, extra_flags=()):
    lib = os.path.abspath(out) if out else LIB
    if not (force or _stale(lib)):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    # objects of a variant build go to a scratch directory so that parallel builds never share files
    objdir = CSRC if out is None else tempfile.mkdtemp(prefix="glomseg_obj_")
    objs = []
    procs = []
    for s in _sources():
        o = os.path.join(objdir, s.replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + list(extra_flags) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    # rpath is only a fallback: inside a torch process the already-loaded libamdhip64.so.7 wins
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        extra = argv[argv.index("--") + 1:]
        argv = argv[:argv.index("--")]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    print(build_lib(force="--force" in argv or out is not None, verbose=True, out=out, extra_flags=extra))
