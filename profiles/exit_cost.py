"""What a process pays at exit for the device memory it still holds: wall time of a child that allocates X GB of HBM
(hipMalloc + hipMemset), then _exit(0)s, for X = 0, 4, 7.2, 14.4.  usage (GPU box): python profiles/exit_cost.py"""
import subprocess
import sys
import time

CHILD = r'''
import ctypes as C, os, sys, time
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
gb = float(sys.argv[1])
ptrs = []
assert hip.hipSetDevice(0) == 0
t1 = time.perf_counter()
n = int(gb * (1 << 30))
if n:
    for _ in range(2):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), n // 2) == 0
        assert hip.hipMemset(p, 0, n // 2) == 0
        ptrs.append(p)
assert hip.hipDeviceSynchronize() == 0
t2 = time.perf_counter()
sys.stderr.write("init %.0f ms alloc+fill %.0f ms\n" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
sys.stderr.flush()
os._exit(0)
'''
for gb in (0, 4, 7.2, 14.4, 0, 14.4):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-c", CHILD, str(gb)], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, r.stderr.strip())
    print("%5.1f GB: wall %.0f ms  (%s)" % (gb, best[0] * 1e3, best[1]))
