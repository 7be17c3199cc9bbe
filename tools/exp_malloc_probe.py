#!/usr/bin/env python3
"""What a fresh 6 GB device allocation costs (on the GPU box): hipMalloc / first touch (memset + sync) / hipFree, three
times over, then again with torch holding 6 GB.  Measured on an MI355X box of this pool: malloc and free 0.3 ms each;
the FIRST touch of a new allocation 0.16 s (the driver clears the pages), 1 ms once the process has touched that much
memory before.  This is the 0.2 s per buffer by which the bench's build times differ from run to run."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipSetDevice(0)
hip.hipDeviceSynchronize()
def t(f):
    t0 = time.perf_counter(); r = f(); return time.perf_counter() - t0, r
out = []
for rep in range(3):
    p = ctypes.c_void_p()
    a, _ = t(lambda: hip.hipMalloc(ctypes.byref(p), 6 << 30))
    m, _ = t(lambda: (hip.hipMemset(p, 0, 6 << 30), hip.hipDeviceSynchronize()))
    f, _ = t(lambda: hip.hipFree(p))
    out.append((round(a, 4), round(m, 4), round(f, 4)))
print("6 GB malloc / memset+sync / free secs:", out)
import torch
x = torch.empty(6 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
out = []
for rep in range(2):
    p = ctypes.c_void_p()
    a, _ = t(lambda: hip.hipMalloc(ctypes.byref(p), 6 << 30))
    f, _ = t(lambda: hip.hipFree(p))
    out.append((round(a, 4), round(f, 4)))
print("with torch holding 6 GB: malloc / free:", out)
