"""Probe of hipIpcOpenMemHandle under the HIP runtime a process ends up with: `import torch` first binds this library to
the runtime bundled with torch (7.0 here), without torch the system runtime (/opt/rocm, 7.2) is used.
RANK / WORLD_SIZE / PROBE_VARIANT (none | torch | torch_import_only | lib_first) / PROBE_N / PROBE_DIR (blob exchange by files)"""
import ctypes, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
variant = os.environ.get("PROBE_VARIANT", "torch")
n = int(os.environ.get("PROBE_N", "30000"))
def log(*a): print(f"[r{rank} {variant} n={n} w={world}]", *a, file=sys.stderr, flush=True)
if variant == "lib_first":
    from dipper_amd import capi
    capi.load_library()
if variant in ("torch", "lib_first"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
    x = torch.randn(512, 512, device="cuda"); y = (x @ x).sum().item()
elif variant == "torch_import_only":
    import torch
import dipper_amd
from dipper_amd import capi as _capi
_capi.load_library()
hip = None
for l in open("/proc/self/maps"):
    p = l.split()[-1]
    if "libamdhip64" in p:
        hip = p; break
v = ctypes.c_int(0); ctypes.CDLL(hip).hipRuntimeGetVersion(ctypes.byref(v))
if rank == 0: log("hip runtime", v.value, hip)
xdir = os.environ["PROBE_DIR"]
d = dipper_amd.Dipper(0); d.set_nj_mode(0)
d.comm_init_local(rank, world)
log("peer_export ..."); blob = d.peer_export(n); log("peer_export done")
open(os.path.join(xdir, "b%d.tmp" % rank), "wb").write(blob)
os.rename(os.path.join(xdir, "b%d.tmp" % rank), os.path.join(xdir, "b%d" % rank))
blobs = []
for r in range(world):
    p = os.path.join(xdir, "b%d" % r)
    while not os.path.exists(p): time.sleep(0.01)
    blobs.append(open(p, "rb").read())
t0 = time.time()
d.peer_attach(blobs); log("peer_attach done in %.2f s" % (time.time() - t0))
open(os.path.join(xdir, "a%d" % rank), "w").close()
for r in range(world):      # nobody frees before everybody has attached
    while not os.path.exists(os.path.join(xdir, "a%d" % r)): time.sleep(0.01)
d.close(); log("ok")
