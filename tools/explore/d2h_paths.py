"""Which engine serves a device -> pinned-host copy on this stack?  Run under rocprofv3 --kernel-trace --memory-copy-trace:
SDMA copies show up in the memory-copy trace, blit copies as __amd_rocclr_copyBuffer kernels.  Variants: torch copy_
(hipMemcpyAsync), hipMemcpy2DAsync, hipMemcpyDtoHAsync, each 16.8 MB."""
import ctypes, sys, time
import torch
torch.cuda.set_device(0)
hip = ctypes.CDLL("libamdhip64.so")
n, h, w = 32, 512, 1024
dev = torch.zeros((n, h, w), dtype=torch.uint8, device="cuda")
host = torch.zeros((n, h, w), dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
torch.cuda.synchronize()
def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print("%-22s %.3f ms  %.1f GB/s" % (name, el * 1e3, dev.numel() / el / 1e9), flush=True)
with torch.cuda.stream(s):
    timed("torch copy_", lambda: host.copy_(dev, non_blocking=True))
sp = ctypes.c_void_p(s.cuda_stream)
hipMemcpyDeviceToHost = 2
def m2d():
    rc = hip.hipMemcpy2DAsync(ctypes.c_void_p(host.data_ptr()), ctypes.c_size_t(h * w), ctypes.c_void_p(dev.data_ptr()), ctypes.c_size_t(h * w),
                              ctypes.c_size_t(h * w), ctypes.c_size_t(n), hipMemcpyDeviceToHost, sp)
    assert rc == 0, rc
timed("hipMemcpy2DAsync", m2d)
def m2d_rows():
    rc = hip.hipMemcpy2DAsync(ctypes.c_void_p(host.data_ptr()), ctypes.c_size_t(w), ctypes.c_void_p(dev.data_ptr()), ctypes.c_size_t(w),
                              ctypes.c_size_t(w), ctypes.c_size_t(n * h), hipMemcpyDeviceToHost, sp)
    assert rc == 0, rc
timed("hipMemcpy2DAsync rows", m2d_rows)
def dtoh():
    rc = hip.hipMemcpyDtoHAsync(ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(dev.data_ptr()), ctypes.c_size_t(dev.numel()), sp)
    assert rc == 0, rc
timed("hipMemcpyDtoHAsync", dtoh)
