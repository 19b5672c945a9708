// What shader clock does the part sustain under a matrix load?  Register-only MFMA loops on every CU (1 or 2 waves per SIMD) that read
// s_memtime (shader clocks) and s_memrealtime (100 MHz reference) around themselves: clock = 100 MHz * d(memtime) / d(memrealtime), next to the
// achieved rate.  The peaks of the guide (157.3 fp32 / 2500 bf16 TFLOP/s) are 2.4 GHz figures; a kernel that keeps the bf16 matrix pipes
// busy runs against the power-managed clock this prints.   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>      // 0: VALU fma only, 1: v_mfma_f32_32x32x2_f32, 2: v_mfma_f32_32x32x16_bf16
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a0) {
    float a = a0 + threadIdx.x * 1e-3f, b = a0 * 0.5f + threadIdx.x * 2e-3f, s = 0;
    bf16x8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (__bf16)(a + e); bv[e] = (__bf16)(b - e); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            else if (MODE == 2) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u & 3], 0, 0, 0);
            else { for (int e = 0; e < 16; ++e) acc[u & 3][e] = __builtin_fmaf(acc[u & 3][e], a, b); }
        }
    }
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = r1 - r0; clk[2 * blockIdx.x + 1] = c1 - c0; }
}
template <int MODE> void run(int wg_per_cu, int iters, const char* what, double flop_per_mfma, double peak) {
    const int blocks = 256 * wg_per_cu;
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned long long* clk; (void)hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {                 // the last repetition is the one reported: clocks have settled by then
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, clk, iters, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int i = 0; i < blocks; ++i) if (h[2 * i]) mhz.push_back(100.0 * (double)h[2 * i + 1] / (double)h[2 * i]);
    std::sort(mhz.begin(), mhz.end());
    const double flops = (double)blocks * 4 * iters * 16 * flop_per_mfma;
    printf("%-28s %d wave(s)/SIMD, %6.2f ms: shader clock %4.0f MHz (min %4.0f, max %4.0f)", what, wg_per_cu, ms, mhz[mhz.size() / 2], mhz.front(), mhz.back());
    if (MODE) printf(";  %7.1f TFLOP/s = %.0f %% of %.1f", flops / ms / 1e9, 100 * flops / ms / 1e9 / peak, peak);
    printf("\n");
    (void)hipFree(out); (void)hipFree(clk);
}
int main() {
    run<0>(1, 20000, "VALU fma only", 0, 0);
    run<1>(1, 40000, "v_mfma_f32_32x32x2_f32", 4096.0, 157.3);
    run<1>(2, 40000, "v_mfma_f32_32x32x2_f32", 4096.0, 157.3);
    run<2>(1, 80000, "v_mfma_f32_32x32x16_bf16", 32768.0, 2500.0);
    run<2>(2, 80000, "v_mfma_f32_32x32x16_bf16", 32768.0, 2500.0);
    run<2>(2, 800000, "v_mfma_f32_32x32x16_bf16", 32768.0, 2500.0);
    return 0;
}
