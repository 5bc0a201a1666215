import ctypes as C, time, sys
sys.path.insert(0, ".")
from dipper_amd import capi
capi.load_library()
import dipper_amd
d = dipper_amd.Dipper(0)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
for rep in range(3):
    p = C.c_void_p()
    t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), 7200000000); t1 = time.perf_counter()
    hip.hipMemset(p, 0, 7200000000); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipFree(p); t3 = time.perf_counter()
    print("malloc %.2f ms  memset %.2f ms  free %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
# one allocation of twice the size vs two (round 3: what the CLI pays between "device ready" and "input done")
for size, k in ((14400000000, 1), (7200000000, 2), (3600000000, 4)):
    t0 = time.perf_counter()
    ps = []
    for i in range(k):
        p = C.c_void_p(); hip.hipMalloc(C.byref(p), size); ps.append(p)
    t1 = time.perf_counter()
    for p in ps: hip.hipFree(p)
    print("%d x %.1f GB: hipMalloc %.2f ms in total" % (k, size / 1e9, (t1 - t0) * 1e3))
