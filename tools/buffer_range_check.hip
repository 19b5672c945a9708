// does the buffer range check include soffset?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* dst, int nrec) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, nrec, 0x00020000);
    // lane t writes dword t at voffset 4t, soffset 256 (64 floats further); num_records = 256 bytes: in range by voffset, out of range by voffset+soffset
    __builtin_amdgcn_raw_buffer_store_b32(0x3f800000u + threadIdx.x, rs, threadIdx.x * 4, 256, 0);
}
int main() {
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 256);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int n = 0; for (int i = 64; i < 128; ++i) n += h[i] != 0.f;
    printf("stores landed with voffset in range and voffset+soffset out of range: %d of 64 -> soffset %s part of the range check\n", n, n ? "is NOT" : "IS");
    return 0;
}
