import ctypes as C, sys
sys.path.insert(0, ".")
import dipper_amd
from dipper_amd import capi
d = dipper_amd.Dipper(0)
L = capi.load_library()
L.dpr_launch_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
for g in (0, 1, 2):
    for grid in (1, 128, 1024, 4096):
        us = C.c_float()
        rc = L.dpr_launch_bench(d.h, 20000, grid, g, C.byref(us))
        if rc != 0:
            L.dpr_last_error.restype = C.c_char_p
            print('mode', g, 'grid', grid, 'failed:', L.dpr_last_error()); continue
        print(("eager", "graph", "graph + new parameters on every node before every replay")[g], "grid", grid, "%.2f us/launch" % us.value)
