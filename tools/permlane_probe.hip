// Semantics of gfx950's v_permlane32_swap / v_permlane16_swap (inline asm: the two-result builtin of this compiler returns the first
// result twice).   hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o /tmp/plp && /tmp/plp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(float* out) {
    float a = (float)threadIdx.x, b = (float)threadIdx.x + 100.f;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
    float c = (float)threadIdx.x, d = (float)threadIdx.x + 100.f;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main() {
    float* d; (void)hipMalloc(&d, 1024); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("inputs: a[l] = l, b[l] = 100 + l\n");
    for (int i = 0; i < 64; i += 8) printf("lane %2d: swap32 -> a' = %3g  b' = %3g     swap16 -> a' = %3g  b' = %3g\n", i, h[i], h[64 + i], h[128 + i], h[192 + i]);
    return 0;
}
