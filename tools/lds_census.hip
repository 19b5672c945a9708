// Census (GPU box): how many 256-thread workgroups of a given dynamic-LDS size / register budget are co-resident on one CU of an MI355X?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_census.hip -o /tmp/lds_census && /tmp/lds_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

template <int NREG>
__global__ __launch_bounds__(256) void census(unsigned long long* out, int spin_us) {
    extern __shared__ float smem[];
    float acc[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) acc[i] = threadIdx.x * 0.5f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) smem[0] = 1.f;
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) acc[i] = acc[i] * 1.0001f + smem[0];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += acc[i];
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x + 0] = t0;
        out[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        out[3 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                  ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) << 32) | ((unsigned long long)(s != 12345.f) << 63);
    }
}

template <int NREG>
void run(int lds_kb) {
    const int nb = 256 * 10;
    unsigned long long* d;
    hipMalloc(&d, nb * 3 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&census<NREG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipLaunchKernelGGL(census<NREG>, dim3(nb), dim3(256), lds_kb * 1024, 0, d, 200);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 3);
    hipMemcpy(h.data(), d, nb * 3 * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int b = 0; b < nb; ++b) {
        unsigned long long hw = h[3 * b + 2] & 0x7fffffffffffffffull;
        unsigned long long cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | ((hw >> 32) << 8);
        ev[cu].push_back({h[3 * b + 0], +1});
        ev[cu].push_back({h[3 * b + 1], -1});
    }
    int best = 0;
    for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0;
        for (auto& p : kv.second) { cur += p.second; best = std::max(best, cur); }
    }
    int numRegs = 0;
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&census<NREG>));
    printf("LDS %3d KB/WG, %3d regs: max co-resident workgroups on a CU = %d  (CUs seen %zu, err %d)\n", lds_kb, fa.numRegs, best, ev.size(), (int)e);
    hipFree(d);
}

int main() {
    for (int kb : {8, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160}) run<8>(kb);
    for (int kb : {8, 32}) { run<40>(kb); run<72>(kb); run<100>(kb); }
    return 0;
}
