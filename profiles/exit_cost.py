"""What the NEXT process pays for the device memory the previous one still held at exit: a child allocates X GB of HBM
(hipMalloc + hipMemset) and _exit(0)s; right after it a second child times its own HIP start-up (runtime init + a stream).
usage (GPU box): python profiles/exit_cost.py"""
import subprocess
import sys
import time

HOLD = r'''
import ctypes as C, os, sys
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
gb = float(sys.argv[1])
assert hip.hipSetDevice(0) == 0
n = int(gb * (1 << 30))
if n:
    for _ in range(2):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), n // 2) == 0
        assert hip.hipMemset(p, 0, n // 2) == 0
assert hip.hipDeviceSynchronize() == 0
os._exit(0)
'''
INIT = r'''
import ctypes as C, os, sys, time
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
cnt = C.c_int()
assert hip.hipGetDeviceCount(C.byref(cnt)) == 0
t1 = time.perf_counter()
s = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
t2 = time.perf_counter()
print("runtime init %.0f ms, stream %.0f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
os._exit(0)
'''
for gb in (0, 2, 7.2, 14.4, 0, 14.4, 28.8):
    res = []
    for _ in range(3):
        subprocess.run([sys.executable, "-c", HOLD, str(gb)], check=True)
        r = subprocess.run([sys.executable, "-c", INIT], capture_output=True, text=True)
        res.append(r.stdout.strip())
    print("previous process held %5.1f GB -> next start-up: %s" % (gb, " | ".join(res)))
