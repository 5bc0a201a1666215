// ipc_probe -- feasibility probe for the peer-window exchange of the row-sharded NJ (profiling aid, not product):
// two PROCESSES on one GPU (forked before any HIP call) share a fine-grained window and a plain hipMalloc buffer through
// hipIpc handles; their kernels ping-pong a sequence number through the windows with bounded spins.  Reports whether the
// two processes' kernels run concurrently, the flag round trip, and whether a peer read of the plain buffer sees the
// data written by the other process's previous kernel.  Last: does RCCL accept two ranks on one device?
//   hipcc --offload-arch=gfx950 -O2 -o bin/ipc_probe ipc_probe.hip -ldl
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x)                                                                                                \
    do {                                                                                                     \
        hipError_t e__ = (x);                                                                                \
        if (e__ != hipSuccess) { std::fprintf(stderr, "[%d] %s -> %s\n", g_rank, #x, hipGetErrorString(e__)); std::exit(3); } \
    } while (0)
static int g_rank = 0;

struct Window { unsigned long long seq[8]; unsigned long long stamp[8]; };

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// rank 0 sends odd numbers into the peer's window and waits for the even answers in its own; rank 1 the other way round
__global__ void pingpong_kernel(Window* mine, Window* peer, int rank, int rounds, unsigned long long timeout_ticks, unsigned long long* out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    unsigned long long ok = 1;
    for (int r = 0; r < rounds && ok; ++r) {
        const unsigned long long want = 2ull * (unsigned long long)r + (rank == 0 ? 2 : 1);
        if (rank == 0) st_sys(&peer->seq[0], 2ull * (unsigned long long)r + 1);
        const unsigned long long ts = wall_clock64();
        while (ld_sys(&mine->seq[0]) < want) {
            if (wall_clock64() - ts > timeout_ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (ok && rank == 1) st_sys(&peer->seq[0], want + 1);
    }
    out[0] = ok;
    out[1] = wall_clock64() - t0;
}

__global__ void fill_kernel(double* buf, int n, double v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) buf[i] = v + i; }
__global__ void check_kernel(const double* buf, int n, double v, unsigned long long* bad, int sysload)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = sysload ? __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : buf[i];
    if (x != v + i) atomicAdd(bad, 1ull);
}

static void xwrite(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) { perror("write"); std::exit(4); } }
static void xread(int fd, void* p, size_t n)
{
    size_t got = 0;
    while (got < n) { const ssize_t r = read(fd, (char*)p + got, n - got); if (r <= 0) { perror("read"); std::exit(4); } got += (size_t)r; }
}

struct Id128 { char b[128]; };

int main()
{
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) return 2;
    const pid_t pid = fork();      // BEFORE any HIP call
    g_rank = pid == 0 ? 1 : 0;
    const int rfd = g_rank == 0 ? b2a[0] : a2b[0], wfd = g_rank == 0 ? a2b[1] : b2a[1];
    alarm(120);
    CK(hipSetDevice(0));
    Window* win = nullptr;
    double* buf = nullptr;
    const int n = 1 << 18;
    hipError_t fe = hipExtMallocWithFlags((void**)&win, 4096, hipDeviceMallocFinegrained);
    std::printf("[%d] hipExtMallocWithFlags(finegrained): %s\n", g_rank, hipGetErrorString(fe));
    if (fe != hipSuccess) CK(hipMalloc((void**)&win, 4096));
    CK(hipMemset(win, 0, 4096));
    CK(hipMalloc((void**)&buf, sizeof(double) * n));
    hipIpcMemHandle_t hw, hb, pw, pb;
    hipError_t e1 = hipIpcGetMemHandle(&hw, win), e2 = hipIpcGetMemHandle(&hb, buf);
    std::printf("[%d] hipIpcGetMemHandle window: %s, buffer: %s\n", g_rank, hipGetErrorString(e1), hipGetErrorString(e2));
    if (e1 != hipSuccess || e2 != hipSuccess) return 5;
    xwrite(wfd, &hw, sizeof hw); xwrite(wfd, &hb, sizeof hb);
    xread(rfd, &pw, sizeof pw); xread(rfd, &pb, sizeof pb);
    Window* pwin = nullptr;
    double* pbuf = nullptr;
    e1 = hipIpcOpenMemHandle((void**)&pwin, pw, hipIpcMemLazyEnablePeerAccess);
    e2 = hipIpcOpenMemHandle((void**)&pbuf, pb, hipIpcMemLazyEnablePeerAccess);
    std::printf("[%d] hipIpcOpenMemHandle window: %s, buffer: %s\n", g_rank, hipGetErrorString(e1), hipGetErrorString(e2));
    if (e1 != hipSuccess || e2 != hipSuccess) return 6;
    unsigned long long* out = nullptr;
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(out, 0, 64));
    char tok = 1;
    // ---- 1. plain buffer: my kernel fills MY buffer; after a host handshake the peer's kernel reads it through its mapping
    for (int sysload = 0; sysload < 2; ++sysload) {
        hipLaunchKernelGGL(fill_kernel, dim3(n / 256), dim3(256), 0, 0, buf, n, 1000.0 * (g_rank + 1) + sysload);
        CK(hipDeviceSynchronize());
        xwrite(wfd, &tok, 1); xread(rfd, &tok, 1);
        CK(hipMemset(out, 0, 64));
        hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, 0, pbuf, n, 1000.0 * ((1 - g_rank) + 1) + sysload, out, sysload);
        CK(hipDeviceSynchronize());
        unsigned long long bad = 0;
        CK(hipMemcpy(&bad, out, 8, hipMemcpyDeviceToHost));
        std::printf("[%d] peer read of the other process's hipMalloc buffer (%s loads): %llu of %d wrong\n", g_rank, sysload ? "system-scope" : "plain", bad, n);
        xwrite(wfd, &tok, 1); xread(rfd, &tok, 1);
    }
    // ---- 2. ping-pong through the windows, both kernels resident at once (bounded spins: 2 s)
    for (int rep = 0; rep < 2; ++rep) {
        const int rounds = 2000;
        CK(hipMemset(win, 0, 4096));
        CK(hipDeviceSynchronize());
        xwrite(wfd, &tok, 1); xread(rfd, &tok, 1);
        hipLaunchKernelGGL(pingpong_kernel, dim3(1), dim3(64), 0, 0, win, pwin, g_rank, rounds, 200000000ull, out);
        CK(hipDeviceSynchronize());
        unsigned long long res[2];
        CK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
        std::printf("[%d] ping-pong %d rounds: %s, %.2f us per round trip (100 MHz clock)\n", g_rank, rounds, res[0] ? "completed" : "TIMED OUT",
                    (double)res[1] / 100.0 / rounds);
        xwrite(wfd, &tok, 1); xread(rfd, &tok, 1);
    }
    // ---- 3. RCCL with two ranks on ONE device
    {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { std::printf("[%d] cannot load librccl\n", g_rank); }
        else {
            auto getid = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
            auto init = (int (*)(void**, int, Id128, int))dlsym(lib, "ncclCommInitRank");
            auto errs = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
            Id128 id;
            if (g_rank == 0) { const int r = getid(&id); std::printf("[0] ncclGetUniqueId: %d\n", r); xwrite(wfd, &id, sizeof id); }
            else xread(rfd, &id, sizeof id);
            void* comm = nullptr;
            const int r = init(&comm, 2, id, g_rank);
            std::printf("[%d] ncclCommInitRank(2 ranks, same device): %d (%s)\n", g_rank, r, errs ? errs(r) : "?");
        }
    }
    std::fflush(stdout);
    if (g_rank == 0) { int st = 0; waitpid(pid, &st, 0); }
    _exit(0);
}
