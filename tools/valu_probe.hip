// Profiling aid (not product): issue rate of the integer VALU instructions the MSA pair kernel is made of (v_xor_b32, v_or3_b32,
// v_and_b32, v_bcnt_u32_b32) on gfx950, by waves per SIMD -- is the SIMD 32 lanes wide for them (2 cycles per wave64 instruction,
// as for v_fma_f32) or 16 (4 cycles)?   make -C tools probe2 && tools/bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __global__ void probe(uint32_t* out, int iters)
{
    uint32_t a[8], b[8], acc[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; b[i] = blockIdx.x * 40503u + i * 7; acc[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { a[i] ^= b[i]; b[i] ^= a[i]; a[i] ^= b[i]; b[i] ^= a[i]; }                       // 4 x v_xor
            if (MODE == 1) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(b[i]));
                             asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(b[i])); }
            if (MODE == 2) { asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i])); asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b[i]), "v"(a[i]));
                             asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[i])); asm volatile("v_or3_b32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b[i]), "v"(a[i])); }
            if (MODE == 3) {      // the pair kernel's 7-op body
                uint32_t m;
                asm volatile("v_or3_b32 %0, %1, %2, %3" : "=v"(m) : "v"(a[i] ^ b[i]), "v"(a[i] ^ acc[(i + 1) & 7]), "v"(b[i] ^ acc[(i + 2) & 7]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(m));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[(i + 3) & 7]) : "v"(a[i] & b[i]));
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + b[i] + acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, int ops_per_iter)
{
    uint32_t* d;
    hipMalloc(&d, sizeof(uint32_t) * 256 * 64 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : { 1, 2, 4, 8 }) {                    // waves per SIMD: blocks of 256 threads (4 waves = 1 per SIMD), wps blocks per CU
        const int grid = 256 * wps, iters = 20000;
        probe<MODE><<<grid, 256>>>(d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<MODE><<<grid, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double lane_ops = (double)grid * 256 * iters * ops_per_iter;
        printf("%-28s %d waves/SIMD: %7.2f T lane-ops/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", name, wps, lane_ops / ms / 1e9,
               2.4e9 * (ms * 1e-3) / ((double)iters * ops_per_iter * wps));
    }
    hipFree(d);
}

int main()
{
    run<0>("v_xor_b32 (4 per step x 8)", 32);
    run<1>("v_bcnt_u32_b32", 32);
    run<2>("v_or3_b32", 32);
    run<3>("pair body: 3 xor, or3, and, 2 bcnt", 56);
    return 0;
}
