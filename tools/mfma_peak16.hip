// Register-only loops of v_mfma_f32_16x16x4_f32 (the attention kernel's instruction) vs v_mfma_f32_32x32x2_f32 (the GEMM's): sustained rates
// with 1, 2, 4 independent accumulators per wave and 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak16.hip -o /tmp/p16 && /tmp/p16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int BIG>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    float s = 0;
    if (BIG) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 32; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int BIG> void run(int wg_per_cu) {
    const int blocks = 256 * wg_per_cu, iters = 20000;
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, BIG>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * (BIG ? 16 * 4096.0 : 32 * 2048.0);
    printf("%s, %d accumulator chain(s) per wave, %d wave(s) per SIMD: %6.1f TF/s (%.0f %% of 157.3)\n", BIG ? "32x32x2 " : "16x16x4 ", NACC, wg_per_cu, flops / best / 1e9, flops / best / 1e9 / 1.573);
    (void)hipFree(out);
}
int main() {
    run<1, 0>(1); run<2, 0>(1); run<4, 0>(1); run<8, 0>(1);
    run<1, 0>(2); run<1, 0>(4); run<4, 0>(4); run<8, 0>(4);
    run<1, 1>(1); run<4, 1>(1); run<1, 1>(4); run<4, 1>(4);
    return 0;
}
