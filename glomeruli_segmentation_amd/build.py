"""Builds libglomseg.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m glomeruli_segmentation_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the working tree.  hipcc cross-compiles
gfx950 code objects without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libglomseg.so")
SOURCES = ["espnet.hip", "detect_ops.hip", "contours.cpp"]
HEADERS = ["gs_internal.h", "conv_mfma.h", "espnet_kernels.h", os.path.join("..", "..", "include", "glomseg.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result",
         "-Wno-unused-value"] + os.environ.get("GS_EXTRA_HIPCC_FLAGS", "").split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    if not (force or _stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(o)
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    # rpath is only a fallback: inside a torch process the already-loaded libamdhip64.so.7 wins
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
