// What the matrix cores SUSTAIN on this device with real operands: register-only loops of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3, unit block scales -- the
// instruction of the fp8 GEMMs), v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x16_f16 on RANDOM vs ALL-ZERO operands, 1 and 2 waves per SIMD, long enough
// (~ 100 ms per launch) for the clock to settle.  No memory traffic, no LDS: the only thing between these figures and the guide's peaks is the clock the power
// budget allows (MI355X_MICROARCH.md, DVFS give-back).  In-kernel clock: s_memtime / s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_fp8.hip -o /tmp/pk8 && /tmp/pk8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int KIND>      // 0 fp8 scaled 32x32x64, 1 bf16 32x32x16, 2 f16 32x32x16
__global__ __launch_bounds__(256) void k(const int* __restrict__ ops, float* out, unsigned long long* clk, int iters) {
    v8i a, b;
    const int* p = ops + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16;
    for (int i = 0; i < 8; ++i) { a[i] = p[i]; b[i] = p[8 + i]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (KIND == 0) acc[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[u & 3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            else if constexpr (KIND == 1) {
                bf16x8 x, y; __builtin_memcpy(&x, &a, 16); __builtin_memcpy(&y, &b, 16);
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[u & 3], 0, 0, 0);
            } else {
                h16x8 x, y; __builtin_memcpy(&x, &a, 16); __builtin_memcpy(&y, &b, 16);
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[u & 3], 0, 0, 0);
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND> void run(const char* name, int wg_per_cu, bool zero, double flop_per_mfma, double peak_tf) {
    const int blocks = 256 * wg_per_cu;
    const size_t n = (size_t)blocks * 256 * 16;
    int* h = (int*)malloc(n * 4);
    srand(7);
    for (size_t i = 0; i < n; ++i) {
        if (zero) h[i] = 0;
        else if (KIND == 0) {                       // four e4m3 bytes of moderate magnitude (exponents 5 .. 9 of 15: |v| in [2^-2, 2^3)), random mantissas and signs
            unsigned v = 0;
            for (int b = 0; b < 4; ++b) v |= (unsigned)((rand() & 0x87) | ((5 + rand() % 5) << 3)) << (8 * b);
            h[i] = (int)v;
        } else {                                    // two 16-bit floats in [0.25, 8) with random mantissas and signs
            unsigned v = 0;
            for (int b = 0; b < 2; ++b) {
                const unsigned e = KIND == 1 ? 125 + rand() % 5 : 13 + rand() % 5, m = rand() & (KIND == 1 ? 0x7f : 0x3ff), sg = rand() & 1;
                v |= (KIND == 1 ? (sg << 15 | e << 7 | m) : (sg << 15 | e << 10 | m)) << (16 * b);
            }
            h[i] = (int)v;
        }
    }
    int* d; float* out; unsigned long long* clk;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&clk, blocks * 16);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    const int iters = KIND == 0 ? 60000 : 120000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, d, out, clk, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long* hc = (unsigned long long*)malloc(blocks * 16);
    (void)hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
    double mhz = 0; for (int b = 0; b < blocks; ++b) mhz += 100.0 * hc[2 * b] / (double)hc[2 * b + 1];
    mhz /= blocks;
    const double tf = (double)blocks * 4 * iters * 16 * flop_per_mfma / ms / 1e9;
    printf("%-28s %s operands, %d wave(s) per SIMD: %7.1f TFLOP/s = %.2f of %.0f; in-kernel clock %4.0f MHz; %.1f ms\n", name, zero ? "ZERO  " : "RANDOM", wg_per_cu, tf, tf / peak_tf, peak_tf, mhz, ms);
    (void)hipFree(d); (void)hipFree(out); (void)hipFree(clk); free(h); free(hc);
}
int main() {
    for (int z = 1; z >= 0; --z)
        for (int w = 1; w <= 2; ++w) {
            run<0>("fp8 scaled 32x32x64", w, z, 2.0 * 32 * 32 * 64, 5000);
            run<1>("bf16 32x32x16", w, z, 2.0 * 32 * 32 * 16, 2500);
            run<2>("f16 32x32x16", w, z, 2.0 * 32 * 32 * 16, 2500);
        }
    return 0;
}
