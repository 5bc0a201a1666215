"""What delays the HIP start-up of a `dipper` run that starts right behind another process?  A predecessor process allocates and
touches `gb` GB of device memory and leaves in one of several ways; the next process's `Device ready in:` line is read.
  python3 profiles/startup_after_teardown.py [gb 15] [repeats 6]
variants of the predecessor: none (no predecessor), small (64 MB), free_exit (hipFree, then exit at once: what the command does),
free_wait (hipFree, 300 ms of sleep, exit), leak_exit (exit without hipFree)"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tmp = tempfile.mkdtemp(prefix="sat_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fa = os.path.join(tmp, "a.fa")
subprocess.run([os.path.join(ROOT, "tools", "bin", "gen_synth"), "--tips", "300", "--sites", "2000", "--seed", "1", "--fasta", fa], check=True)
pred = r"""
import ctypes as C, sys, time
mode, nbytes = sys.argv[1], int(float(sys.argv[2]))
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
ptrs = []
left = nbytes
while left > 0:
    n = min(left, 4 << 30); p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n) == 0 and hip.hipMemset(p, 1, n) == 0
    ptrs.append(p); left -= n
assert hip.hipDeviceSynchronize() == 0
if mode != "leak_exit":
    for p in ptrs: assert hip.hipFree(p) == 0
if mode == "free_wait": time.sleep(0.3)
import os; os._exit(0)
"""
exe = os.path.join(ROOT, "dipper_amd", "bin", "dipper")
def ready_ms():
    r = subprocess.run([exe, "-i", "m", "-I", fa, "-O", os.path.join(tmp, "o.nwk"), "-m", "2", "-d", "2"], capture_output=True, text=True)
    for line in r.stderr.splitlines():
        if line.startswith("Device ready in:"):
            return float(line.split(":")[1].split()[0])
    return None
out = {}
for mode, nbytes in (("none", 0), ("small", 64e6), ("free_exit", gb * 1e9), ("free_wait", gb * 1e9), ("leak_exit", gb * 1e9), ("none", 0)):
    vals = []
    for _ in range(reps):
        if mode == "none":
            time.sleep(0.5)
        else:
            subprocess.run([sys.executable, "-c", pred, mode, str(nbytes)], check=True)
        vals.append(ready_ms())
    out.setdefault(mode, []).append(vals)
    print(mode, vals, flush=True)
print(json.dumps({"gb": gb, "device_ready_ms": out}))
