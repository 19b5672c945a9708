// Micro-benchmark 2 (GPU box): WHY does the LDS-DMA operand stream cost ~12 % of the MFMA rate in the fp32 GEMM loop?
//   hipcc --offload-arch=gfx950 -O3 tools/loop_bench2.hip -o /tmp/loop_bench2 && /tmp/loop_bench2
// Same loop body as loop_bench.hip (4 MFMA waves, 64x64 per wave, 32 x v_mfma_f32_32x32x2_f32 + 8 ds_read_b128 + 1 barrier per K step).
// MODE: 0 no operand stream; 1 LDS-DMA by the MFMA waves (production); 2 LDS-DMA by the MFMA waves but NO fragment reads;
//       3 global_load_dwordx4 into registers only (no LDS write); 4 a 5th PRODUCER wave issues all 16 pieces, MFMA waves issue none;
//       5 like 1 with dword-sized DMA (4x the instructions, same bytes); 6 producer wave AND no fragment reads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int NBUF, int NDMA>
__global__ __launch_bounds__(320) void loop(const float* __restrict__ src, size_t src_floats, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = 256 * 16;
    constexpr bool READS = MODE != 2 && MODE != 6, PROD = MODE == 4 || MODE == 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < NBUF * STAGE; i += blockDim.x) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    size_t pos = ((size_t)blockIdx.x * 4099 * 4096) % src_floats;
    if (PROD && wave == 4) {
        const float* sp = src + pos + lane * 4;
        int stg = 0;
        for (int kt = 0; kt < steps + NBUF - 1; ++kt) {
#pragma unroll
            for (int u = 0; u < 4 * NDMA; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(sp + u * 256), (lptr_t)(smem + stg * STAGE + u * 256), 16, 0, 0);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
            stg = stg + 1 == NBUF ? 0 : stg + 1;
            if (kt >= NBUF - 2) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * 4 * NDMA) : "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 a0[2], b0[2], a1[2], b1[2];
    for (int i = 0; i < 2; ++i) { a0[i] = f32x4{0.5f + lane * 1e-3f, 0.25f, 0.125f, 1.f}; b0[i] = f32x4{0.3f, 0.7f + lane * 1e-3f, 0.2f, 0.9f}; a1[i] = b0[i]; b1[i] = a0[i]; }
    const int a_row = ((wave >> 1) * 64 + l31) * 16, b_row = (128 + (wave & 1) * 64 + l31) * 16;
    const float* sp = src + pos + (size_t)wave * 4 * 256 + lane * 4;
    int cur = 0, stg = NBUF - 1;
    auto rd = [&](int buf, int g, f32x4 (&af)[2], f32x4 (&bf)[2]) {
        const int cg = 4 * (2 * g + lh);
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + a_row + i * 32 * 16 + cg);
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + b_row + j * 32 * 16 + cg);
    };
    auto mm = [&](const f32x4 (&af)[2], const f32x4 (&bf)[2]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
    };
    constexpr int NV = MODE == 5 ? 4 * NDMA : NDMA;          // VMEM instructions per wave-step
    if (MODE == 1 || MODE == 2 || MODE == 5) {
        for (int t = 0; t < NBUF - 1; ++t) {
            for (int u = 0; u < 4; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(sp + u * 256), (lptr_t)(smem + t * STAGE + (wave * 4 + u) * 256), 16, 0, 0);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (PROD) __builtin_amdgcn_s_barrier();
    f32x4 stg_r[NDMA];
    for (int kt = 0; kt < steps; ++kt) {
        const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
        if (READS) rd(cur, 1, a1, b1);
        if (MODE == 1 || MODE == 2) {
            for (int u = 0; u < NDMA; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(sp + u * 256), (lptr_t)(smem + stg * STAGE + (wave * 4 + u) * 256), 16, 0, 0);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        if (MODE == 5) {
            const float* s1 = sp - lane * 4 + lane;                  // 64 lanes x 4 B = 256 B per instruction
            for (int u = 0; u < 4 * NDMA; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(s1 + u * 64), (lptr_t)(smem + stg * STAGE + wave * 1024 + u * 64), 4, 0, 0);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        if (MODE == 3) {
            if (kt) for (int u = 0; u < NDMA; ++u) asm volatile("" ::"v"(stg_r[u]));
            for (int u = 0; u < NDMA; ++u) stg_r[u] = *reinterpret_cast<const f32x4*>(sp + u * 256);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        mm(a0, b0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (READS) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        if (MODE == 1 || MODE == 2 || MODE == 5) for (int u = 0; u < NV && u < 14; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
        if (MODE == 3) for (int u = 0; u < NDMA; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2 || MODE == 5) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NV) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (READS) rd(nxt, 0, a0, b0);
        mm(a1, b1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        if (READS) __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 15, 1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt; stg = stg + 1 == NBUF ? 0 : stg + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NBUF, int NDMA>
void run(const char* what, int wg_per_cu, const float* src, size_t src_mb) {
    const int blocks = 256 * wg_per_cu, steps = 4096;
    const int threads = (MODE == 4 || MODE == 6) ? 320 : 256;
    const int lds = wg_per_cu == 1 ? 150 * 1024 : (wg_per_cu == 2 ? 80 * 1024 : NBUF * 256 * 16 * 4);      // padding forces the residency
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<MODE, NBUF, NDMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<MODE, NBUF, NDMA>), dim3(blocks), dim3(threads), lds, 0, src, src_mb * 1024 * 1024 / 4, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    const double flops = (double)blocks * 4 * steps * 32 * 4096.0;
    printf("%-72s WG/CU %d stages %d: %7.2f ms %6.1f TF/s (%.0f %%)%s\n", what, wg_per_cu, NBUF, best, flops / best / 1e9, flops / best / 1e9 / 1.573, e == hipSuccess ? "" : hipGetErrorString(e));
    hipFree(out);
}

int main() {
    float* src; const size_t big = 1024;
    hipMalloc(&src, big * 1024 * 1024);
    hipMemset(src, 0x3c, big * 1024 * 1024);
    for (int wg = 2; wg >= 1; --wg) {
        if (wg == 2) {
            run<0, 4, 4>("0: MFMA + fragment reads + barrier", 2, src, big);
            run<1, 4, 4>("1: + LDS-DMA x4 per wave-step by the MFMA waves (production)", 2, src, big);
            run<2, 4, 4>("2: LDS-DMA x4, NO fragment reads", 2, src, big);
            run<3, 4, 4>("3: global_load_dwordx4 x4 into registers only (no LDS write)", 2, src, big);
            run<4, 4, 4>("4: producer wave issues 16 pieces per step, MFMA waves none", 2, src, big);
            run<6, 4, 4>("6: producer wave, NO fragment reads", 2, src, big);
            run<5, 4, 4>("5: dword LDS-DMA x16 per wave-step (same bytes, 4x instructions)", 2, src, big);
            run<1, 4, 2>("1: LDS-DMA x2", 2, src, big);
            run<4, 4, 2>("4: producer wave, 8 pieces per step", 2, src, big);
            run<1, 4, 4>("1: LDS-DMA x4, source = 1 MB (L2-resident)", 2, src, 1);
            run<4, 4, 4>("4: producer wave, source = 1 MB", 2, src, 1);
        } else {
            run<0, 4, 4>("0: MFMA + fragment reads + barrier", 1, src, big);
            run<1, 4, 4>("1: LDS-DMA x4 by the MFMA waves", 1, src, big);
            run<4, 4, 4>("4: producer wave", 1, src, big);
        }
    }
    return 0;
}
