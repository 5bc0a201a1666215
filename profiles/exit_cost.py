"""What does the END of a process cost that holds device memory?  python profiles/exit_cost.py
child: hipMalloc + hipMemset of X GB in k buffers, hipDeviceSynchronize, os._exit(0); parent: wall of the child minus the child's own clock."""
import ctypes, os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    gb, k, free = float(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1"
    t0 = time.perf_counter()
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    ptrs = []
    for i in range(k):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gb * (1 << 30) / k))) == 0
        assert hip.hipMemset(p, 1, ctypes.c_size_t(int(gb * (1 << 30) / k))) == 0
        ptrs.append(p)
    hip.hipDeviceSynchronize()
    if free:
        for p in ptrs:
            hip.hipFree(p)
    sys.stderr.write("CHILD %.1f\n" % ((time.perf_counter() - t0) * 1e3))
    sys.stderr.flush()
    os._exit(0)
for gb, k, free in ((0.1, 1, 0), (2, 2, 0), (8, 2, 0), (16, 2, 0), (16, 2, 1), (16, 16, 0), (0.1, 1, 0)):
    for r in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(gb), str(k), str(free)], capture_output=True, text=True)
        wall = (time.perf_counter() - t0) * 1e3
        child = [float(l.split()[1]) for l in p.stderr.splitlines() if l.startswith("CHILD")]
        print(f"{gb:5.1f} GB in {k:2d} buffers, hipFree before exit {free}: wall {wall:7.1f} ms, child's own clock {child[0] if child else -1:7.1f} ms, outside {wall - (child[0] if child else 0):6.1f} ms", flush=True)
