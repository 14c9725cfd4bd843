"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST-side native code of the product (sanitizers on
the CPU build only; GPU sanitizers are not available on this pool): csrc/contours.cpp (Suzuki-Abe border following, arc length,
Douglas-Peucker) is compiled on its own with g++ -fsanitize=address,undefined and driven through its C ABI from a child
process that preloads the sanitizer runtime."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

from conftest import REPO

STUB = r"""
#include <cstdarg>
#include <cstdio>
namespace gs { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); std::vfprintf(stderr, fmt, ap); va_end(ap); std::fputc('\n', stderr); } }
"""

CHILD = r"""
import ctypes, sys
import numpy as np
lib = ctypes.CDLL(sys.argv[1])
lib.gs_find_contours.restype = ctypes.c_int
lib.gs_arc_length_closed.restype = ctypes.c_double
lib.gs_approx_poly_closed.restype = ctypes.c_int
rng = np.random.default_rng(5)
total = 0
for trial in range(12):
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    img = (rng.random((h, w)) < (0.15 + 0.07 * trial)).astype(np.uint8)
    if trial % 3 == 0:                       # blobs that touch every border
        img[0, :] = 1; img[-1, :] = 1; img[:, 0] = 1; img[:, -1] = 1
    for simple in (0, 1):
        nc, npnt = ctypes.c_int(0), ctypes.c_int(0)
        rc = lib.gs_find_contours(img.ctypes.data_as(ctypes.c_void_p), h, w, simple, None, 0, None, 0, ctypes.byref(nc), ctypes.byref(npnt))
        assert rc == 0, rc
        pts = np.zeros((max(npnt.value, 1), 2), dtype=np.int32)
        offs = np.zeros(nc.value + 1, dtype=np.int32)
        rc = lib.gs_find_contours(img.ctypes.data_as(ctypes.c_void_p), h, w, simple, pts.ctypes.data_as(ctypes.c_void_p), npnt.value,
                                  offs.ctypes.data_as(ctypes.c_void_p), nc.value + 1, ctypes.byref(nc), ctypes.byref(npnt))
        assert rc == 0, rc
        assert offs[0] == 0 and offs[-1] == npnt.value
        # one point too few: must fail cleanly, not write past the buffer
        if npnt.value > 1:
            rc2 = lib.gs_find_contours(img.ctypes.data_as(ctypes.c_void_p), h, w, simple, pts.ctypes.data_as(ctypes.c_void_p), npnt.value - 1,
                                       offs.ctypes.data_as(ctypes.c_void_p), nc.value + 1, ctypes.byref(ctypes.c_int(0)), ctypes.byref(ctypes.c_int(0)))
            assert rc2 != 0
        for k in range(nc.value):
            c = np.ascontiguousarray(pts[offs[k]:offs[k + 1]])
            if len(c) == 0:
                continue
            length = lib.gs_arc_length_closed(c.ctypes.data_as(ctypes.c_void_p), len(c))
            out = np.zeros_like(c)
            m = lib.gs_approx_poly_closed(c.ctypes.data_as(ctypes.c_void_p), len(c), ctypes.c_double(0.01 * length + 0.5), out.ctypes.data_as(ctypes.c_void_p))
            assert 0 <= m <= len(c)
            total += m
print("points kept", total)
"""


def test_contours_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not gxx or not os.path.isabs(asan) or not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("g++ / libasan / HIP headers not available")
    stub = tmp_path / "stub.cpp"
    stub.write_text(STUB)
    so = tmp_path / "libcontours_asan.so"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(REPO, "glomeruli_segmentation_amd", "csrc", "contours.cpp"),
           str(stub), "-o", str(so)]
    subprocess.check_call(cmd)
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(CHILD))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77", UBSAN_OPTIONS="halt_on_error=1:exitcode=78")
    res = subprocess.run([sys.executable, str(child), str(so)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.returncode, res.stderr[-3000:])
    assert "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-3000:]
    assert "points kept" in res.stdout


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_crop_pipeline_host_side_under_sanitizers(tmp_path, san):
    """the host code of gs_espnet_segment_crops_host that has no GPU in it -- the batch planner and packed-slot layout
    (csrc/crop_plan.h) and the threaded staging copies (csrc/host_jobs.h) -- replayed by tests/helpers/host_side_driver.cpp on
    1 000 random crop lists (lengths 0, 1, 7, 8, 63, 64, 65, ... and a list with 4000 x 7000 crops) under
    AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer"""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = tmp_path / "host_side_driver"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + san, "-fno-sanitize-recover=all",
           "-I" + os.path.join(REPO, "glomeruli_segmentation_amd", "csrc"),
           os.path.join(REPO, "tests", "helpers", "host_side_driver.cpp"), "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr and ("cannot find" in built.stderr or "unrecognized" in built.stderr):
        pytest.skip("sanitizer runtime not available: " + built.stderr[-300:])
    assert built.returncode == 0, built.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=77", UBSAN_OPTIONS="halt_on_error=1:exitcode=78",
               TSAN_OPTIONS="halt_on_error=1:exitcode=79")
    res = subprocess.run([str(exe), "1000"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-3000:])
    assert "host side ok: 1" in res.stdout
    for word in ("AddressSanitizer", "ThreadSanitizer", "runtime error"):
        assert word not in res.stderr, res.stderr[-3000:]
