// rowops.hip's LDS-free wave reductions (the MMDM_ROWOPS_NOPK build: DPP inside a 16-lane row + v_permlane16/32_swap across rows) against the
// __shfl_xor butterflies they replace: every lane must hold the wave's sum (to rounding) and EXACTLY the wave's maximum.
//   hipcc --offload-arch=gfx950 -O3 tools/wave_reduce_probe.hip -o /tmp/wrp && /tmp/wrp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = a + b; b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp<0xB1>(v)); v = fmaxf(v, dpp<0x4E>(v)); v = fmaxf(v, dpp<0x141>(v)); v = fmaxf(v, dpp<0x140>(v));
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b); b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__global__ void k(const float* in, float* s, float* m, float* s0, float* m0) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float v = in[i];
    s[i] = wave_sum(v); m[i] = wave_max(v);
    float a = v, b = v;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b = fmaxf(b, __shfl_xor(b, o)); }
    s0[i] = a; m0[i] = b;
}
int main() {
    const int W = 4096, n = W * 64;
    float* h = (float*)malloc(n * 4); srand(3);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 8.f - 4.f;
    float *d, *s, *m, *s0, *m0; (void)hipMalloc(&d, n * 4); (void)hipMalloc(&s, n * 4); (void)hipMalloc(&m, n * 4); (void)hipMalloc(&s0, n * 4); (void)hipMalloc(&m0, n * 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, d, s, m, s0, m0);
    float *hs = (float*)malloc(n * 4), *hm = (float*)malloc(n * 4), *hm0 = (float*)malloc(n * 4);
    (void)hipMemcpy(hs, s, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hm, m, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hm0, m0, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int w = 0; w < W; ++w) {
        double ref = 0; float mx = -1e30f;
        for (int l = 0; l < 64; ++l) { ref += h[w * 64 + l]; mx = fmaxf(mx, h[w * 64 + l]); }
        for (int l = 0; l < 64; ++l) {
            const double e = fabs(hs[w * 64 + l] - ref); if (e > worst) worst = e;
            if (e > 1e-4 || hm[w * 64 + l] != mx || hm0[w * 64 + l] != mx) ++bad;
        }
    }
    printf("wave reductions: %d lanes wrong of %d, largest |sum - float64 sum| %.2e\n", bad, n, worst);
    return bad != 0;
}
