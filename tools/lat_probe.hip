// lat_probe -- memory round-trip times on one MI355X as a single wave sees them (profiling aid, not product).
// The kernels of the pruned NJ loop and of the placement loop are chains of dependent memory operations (DESIGN.md
// section 4); what one link of such a chain costs decides which restructurings can pay at all.  One thread, the 100 MHz
// wall clock (10 ns steps), many repetitions:
//   load  : dependent pointer chase over a working set of 16 KB / 2 MB / 64 MB / 4 GB (L1-L2 / L2 / Infinity Cache / HBM,
//           the last one with a page step so that every access also misses the TLB), plain and system-scope (sc0 sc1) loads
//   store : one 8-byte store + s_waitcnt vmcnt(0): plain, agent-scope write-through (the new node's column in njp.hip),
//           non-temporal; to a hot line and to a fresh line of a 4 GB buffer
//   atomic: atomicAdd that returns (agent scope) on a hot word -- the list append of the post kernels, a ticket
//   chain : launch-to-launch time of an empty kernel chain is measured by dpr_launch_bench (library), not here.
//   hipcc --offload-arch=gfx950 -O2 -o bin/lat_probe lat_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x)                                                                                              \
    do {                                                                                                   \
        hipError_t e__ = (x);                                                                              \
        if (e__ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e__)); std::exit(3); } \
    } while (0)

template <int MODE> __device__ __forceinline__ uint64_t ld(const uint64_t* p)
{
    if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == 2) return __builtin_nontemporal_load(p);
    return *p;
}

// chase[i] = index of the next element (in units of uint64_t); `steps` dependent loads after `warm` untimed ones.
// Lane 0 walks the chain with VECTOR loads (the address goes through a VGPR the compiler cannot prove uniform), as the
// kernels of the NJ loop do; a uniform address would become a scalar load through the constant cache.
template <int MODE> __global__ void chase_kernel(const uint64_t* buf, uint64_t start, int warm, int steps, uint64_t* out)
{
    if (threadIdx.x != 0) return;
    uint64_t i = start;
    asm volatile("" : "+v"(i));
    for (int k = 0; k < warm; ++k) i = ld<MODE>(buf + i);
    const uint64_t t0 = wall_clock64();
    for (int k = 0; k < steps; ++k) i = ld<MODE>(buf + i);
    const uint64_t t1 = wall_clock64();
    out[0] = t1 - t0;
    out[1] = i;
}

// MODE 0 plain, 1 agent-scope write-through (relaxed atomic store), 2 non-temporal, 3 system scope
template <int MODE> __global__ void store_kernel(uint64_t* buf, uint64_t stride, int steps, uint64_t* out)
{
    if (threadIdx.x != 0) return;
    uint64_t acc = 0;
    for (int k = 0; k < steps; ++k) {
        uint64_t off = (uint64_t)k * stride;
        asm volatile("" : "+v"(off));
        uint64_t* p = buf + off;
        const uint64_t t0 = wall_clock64();
        if (MODE == 1) __hip_atomic_store(p, (uint64_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) __builtin_nontemporal_store((uint64_t)k, p);
        else if (MODE == 3) __hip_atomic_store(p, (uint64_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *p = (uint64_t)k;
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += wall_clock64() - t0;
    }
    out[0] = acc;
}

__global__ void atomic_kernel(unsigned long long* word, int steps, uint64_t* out)
{
    if (threadIdx.x != 0) return;
    uint64_t acc = 0, sum = 0;
    for (int k = 0; k < steps; ++k) {
        const uint64_t t0 = wall_clock64();
        const unsigned long long v = __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += v;                      // (uses the returned value: the wave waits for it)
        asm volatile("" : "+v"(sum));
        acc += wall_clock64() - t0;
    }
    out[0] = acc;
    out[1] = sum;
}

// the same atomic while `blocks` other workgroups hammer the same word (a list counter that 250 blocks append to)
__global__ void atomic_contended_kernel(unsigned long long* word, int steps, uint64_t* out)
{
    if (threadIdx.x != 0) return;
    uint64_t acc = 0, sum = 0;
    for (int k = 0; k < steps; ++k) {
        const uint64_t t0 = wall_clock64();
        sum += __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" : "+v"(sum));
        acc += wall_clock64() - t0;
    }
    if (blockIdx.x == 0) { out[0] = acc; out[1] = sum; }
}

// (two launches, the first one warms the code path; the second CONTINUES the chain where the first one ended, so a large
//  working set is met cold -- round 4's first version restarted at element 0 and timed 2 000 lines the first launch had
//  just pulled into L2: every row read 95 ns)
static uint64_t g_cursor = 0;
static double run_chase(int mode, const uint64_t* d, int warm, int steps, uint64_t* dout)
{
    uint64_t h[2] = { 0, g_cursor };
    for (int rep = 0; rep < 2; ++rep) {
        const uint64_t start = h[1];
        if (mode == 1) hipLaunchKernelGGL(chase_kernel<1>, dim3(1), dim3(64), 0, 0, d, start, warm, steps, dout);
        else if (mode == 2) hipLaunchKernelGGL(chase_kernel<2>, dim3(1), dim3(64), 0, 0, d, start, warm, steps, dout);
        else hipLaunchKernelGGL(chase_kernel<0>, dim3(1), dim3(64), 0, 0, d, start, warm, steps, dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost));
    }
    g_cursor = h[1];
    return 10.0 * (double)h[0] / steps;      // ns per dependent load
}

// dispatch ramp: every workgroup stamps the clock when it starts (then idles ~3 us so that early workgroups do not make room
// for late ones); the spread of the stamps is how long the dispatcher takes to get a grid going
__global__ void ramp_kernel(uint64_t* stamps)
{
    const uint64_t t0 = wall_clock64();
    if (threadIdx.x == 0) stamps[blockIdx.x] = t0;
    while (wall_clock64() - t0 < 300) { }
}

int main()
{
    CK(hipSetDevice(0));
    uint64_t* dout = nullptr;
    CK(hipMalloc(&dout, 64));
    std::mt19937_64 rng(1);
    std::printf("{\n");
    // ---- loads
    struct WS { const char* name; size_t bytes; size_t step; };     // step: distance between chain elements (bytes)
    // 16 KB / 2 MB: one lap first, then timed (cache hits).  64 MB (Infinity Cache after its own upload), 2 GB with a page
    // step and 8 GB with a 2 MB step (every access a fresh line in a fresh page / fragment): met cold
    const WS sets[] = { { "16KB", 16u << 10, 64 }, { "2MB", 2u << 20, 128 }, { "64MB", 64u << 20, 256 }, { "2GB_page_step", (size_t)2 << 30, 4096 + 256 },
                        { "8GB_2MB_step", (size_t)8 << 30, ((size_t)2 << 20) + 4096 + 256 } };
    for (const WS& w : sets) {
        const size_t nel = w.bytes / w.step;
        std::vector<uint64_t> order(nel);
        std::iota(order.begin(), order.end(), 0);
        std::shuffle(order.begin() + 1, order.end(), rng);
        std::vector<uint64_t> host(w.bytes / 8, 0);
        for (size_t k = 0; k < nel; ++k) host[order[k] * (w.step / 8)] = order[(k + 1) % nel] * (w.step / 8);
        uint64_t* d = nullptr;
        CK(hipMalloc(&d, w.bytes));
        CK(hipMemcpy(d, host.data(), w.bytes, hipMemcpyHostToDevice));
        const bool resident = w.bytes <= (2u << 20);
        const int steps = (int)std::min<size_t>(nel / 8, 1000);    // (three modes x two launches stay on fresh elements)
        const int warm = resident ? (int)nel : 0;                  // small sets: one lap first, so the timed lap hits the cache
        g_cursor = 0;
        const double a0 = run_chase(0, d, warm, resident ? std::min<int>((int)nel, 2000) : steps, dout);
        const double a1 = run_chase(1, d, warm, resident ? std::min<int>((int)nel, 2000) : steps, dout);
        const double a2 = run_chase(2, d, warm, resident ? std::min<int>((int)nel, 2000) : steps, dout);
        std::printf(" \"load_ns_%s\": {\"plain\": %.0f, \"system_scope\": %.0f, \"nontemporal\": %.0f, \"elements\": %zu},\n", w.name, a0, a1, a2, nel);
        CK(hipFree(d));
    }
    // ---- stores
    {
        const size_t big = (size_t)2 << 30;
        uint64_t* d = nullptr;
        CK(hipMalloc(&d, big));
        CK(hipMemset(d, 0, big));
        const int steps = 200;
        uint64_t base_words = 0;                                    // every launch writes lines no earlier launch touched
        auto st = [&](int mode, uint64_t stride) {
            uint64_t h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                uint64_t* b = d + (stride ? base_words : 0);
                if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(1), dim3(64), 0, 0, b, stride, steps, dout);
                else if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(1), dim3(64), 0, 0, b, stride, steps, dout);
                else if (mode == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(1), dim3(64), 0, 0, b, stride, steps, dout);
                else hipLaunchKernelGGL(store_kernel<3>, dim3(1), dim3(64), 0, 0, b, stride, steps, dout);
                CK(hipDeviceSynchronize());
                base_words += 64 + (stride < 1000 ? (uint64_t)steps * stride : 1024);      // (the far stride walks 4 MB steps: shift by a few KB)
            }
            CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
            return 10.0 * (double)h / steps;
        };
        const uint64_t far = ((size_t)4 << 20) / 8 + 72;            // a fresh line (and page) every time: 200 x 4 MB = 800 MB of the 2 GB
        std::printf(" \"store_drain_ns\": {\"same_line\": {\"plain\": %.0f, \"agent_writethrough\": %.0f, \"nontemporal\": %.0f, \"system\": %.0f},\n", st(0, 0), st(1, 0), st(2, 0), st(3, 0));
        std::printf("                    \"fresh_line_4MB_apart\": {\"plain\": %.0f, \"agent_writethrough\": %.0f, \"nontemporal\": %.0f, \"system\": %.0f},\n", st(0, far), st(1, far), st(2, far), st(3, far));
        std::printf("                    \"fresh_line_256B_apart\": {\"plain\": %.0f, \"agent_writethrough\": %.0f, \"nontemporal\": %.0f, \"system\": %.0f}},\n", st(0, 32), st(1, 32), st(2, 32), st(3, 32));
        CK(hipFree(d));
    }
    // ---- atomics
    {
        unsigned long long* w = nullptr;
        CK(hipMalloc(&w, 256));
        CK(hipMemset(w, 0, 256));
        uint64_t h = 0;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(atomic_kernel, dim3(1), dim3(64), 0, 0, w, 1000, dout); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
        std::printf(" \"atomic_add_return_ns\": {\"alone\": %.0f", 10.0 * (double)h / 1000);
        for (int blocks : { 64, 256, 1024 }) {
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(atomic_contended_kernel, dim3(blocks), dim3(64), 0, 0, w, 200, dout); CK(hipDeviceSynchronize()); }
            CK(hipMemcpy(&h, dout, 8, hipMemcpyDeviceToHost));
            std::printf(", \"with_%d_blocks_on_the_word\": %.0f", blocks, 10.0 * (double)h / 200);
        }
        std::printf("},\n");
        CK(hipFree(w));
    }
    // ---- dispatch ramp: last start - first start of a grid, by shape (the post launch of the NJ loop: ~1 000 x 256 threads)
    {
        uint64_t* st = nullptr;
        CK(hipMalloc(&st, 8 * 8192));
        std::printf(" \"dispatch_ramp_ns\": {");
        const int shapes[][2] = { { 256, 256 }, { 512, 256 }, { 1024, 256 }, { 2048, 256 }, { 1024, 128 }, { 2048, 128 }, { 512, 512 }, { 256, 1024 }, { 1024, 64 }, { 4096, 64 } };
        bool first = true;
        for (const auto& sh : shapes) {
            std::vector<uint64_t> h((size_t)sh[0]);
            double best = 1e30, med_sum = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(ramp_kernel, dim3(sh[0]), dim3(sh[1]), 0, 0, st);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), st, 8 * (size_t)sh[0], hipMemcpyDeviceToHost));
                const uint64_t lo = *std::min_element(h.begin(), h.end()), hi = *std::max_element(h.begin(), h.end());
                if (rep) { best = std::min(best, 10.0 * (double)(hi - lo)); med_sum += 10.0 * (double)(hi - lo); }
            }
            std::printf("%s\"%dx%d\": {\"min\": %.0f, \"mean\": %.0f}", first ? "" : ", ", sh[0], sh[1], best, med_sum / 4);
            first = false;
        }
        std::printf("}\n");
        CK(hipFree(st));
    }
    std::printf("}\n");
    return 0;
}
