#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template<int WAVES>
__global__ __launch_bounds__(64*WAVES) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i=0;i<4;++i) for (int e=0;e<16;++e) acc[i][e]=0.f;
    float a = a0 + threadIdx.x*1e-3f, b = b0 + threadIdx.x*2e-3f;
    for (int it=0; it<iters; ++it) {
#pragma unroll
        for (int u=0;u<8;++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0],0,0,0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1],0,0,0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2],0,0,0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3],0,0,0);
        }
    }
    float s=0; for (int i=0;i<4;++i) for (int e=0;e<16;++e) s+=acc[i][e];
    out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int WAVES> void run(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks*64*WAVES*4);
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep=0; rep<3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<WAVES>, dim3(blocks), dim3(64*WAVES), 0, 0, out, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms,e0,e1);
        double flops = (double)blocks*WAVES*iters*32*4096.0;
        printf("waves/WG=%d blocks=%d iters=%d: %.3f ms  %.1f TF/s\n", WAVES, blocks, iters, ms, flops/ms/1e9);
    }
    hipFree(out);
}
int main(){ run<4>(256, 20000); run<4>(512, 20000); run<8>(256,20000); run<4>(256, 200000); return 0; }
