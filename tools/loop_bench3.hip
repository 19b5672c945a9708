// Micro-benchmark 3 (GPU box): the fp32 GEMM K loop at ONE wave per SIMD (one 4-wave workgroup per CU) -- placement / addressing knobs.
//   hipcc --offload-arch=gfx950 -O3 tools/loop_bench3.hip -o /tmp/loop_bench3 && /tmp/loop_bench3
// TM x 2 MFMA tiles of 32x32 per wave (TM = 2: 64x64, TM = 4: 128x64), K step 16 in two k-groups; per step and wave: 16*TM MFMAs,
// 2*(TM+2) ds_read_b128, NDMA LDS-DMA pieces, one barrier.
// ADDR 0: global_load_lds_dwordx4 with per-lane 64-bit pointers (VALU increments); 1: buffer_load_dwordx4 ... offen lds, fixed per-lane
//         offset + SGPR soffset (SALU increments).
// PLACE 0: DMAs one per MFMA behind the first MFMA of k-group 0; 1: burst before the MFMAs of the step; 2: half in group 0, half in group 1.
// BAR 0: no barrier (timing only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int TM, int NBUF, int NDMA, int ADDR, int PLACE, int BAR>
__global__ __launch_bounds__(256) void loop(const float* __restrict__ src, size_t src_floats, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 2 * 32 * TM + 128, STAGE = ROWS * 16;      // A rows (2 waves in M) + 128 B rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < NBUF * STAGE; i += blockDim.x) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    f32x16 acc[TM][2];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 a0[TM], b0[2], a1[TM], b1[2];
    for (int i = 0; i < TM; ++i) { a0[i] = f32x4{0.5f + lane * 1e-3f, 0.25f, 0.125f, 1.f}; a1[i] = f32x4{0.3f, 0.7f + lane * 1e-3f, 0.2f, 0.9f}; }
    for (int j = 0; j < 2; ++j) { b0[j] = a1[0]; b1[j] = a0[0]; }
    const int a_row = ((wave >> 1) * 32 * TM + l31) * 16, b_row = (2 * 32 * TM + (wave & 1) * 64 + l31) * 16;
    size_t pos = ((size_t)blockIdx.x * 4099 * 4096) % src_floats;
    if (pos + (size_t)(steps + NBUF) * 1024 * NDMA + 65536 > src_floats) pos = 0;
    const float* sp = src + pos + (size_t)wave * NDMA * 256 + lane * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + pos), 0, 0x7fffffff, 0x00020000);
    const int voff = (wave * NDMA * 256 + lane * 4) * 4;
    int soff = 0;
    int cur = 0, stg = NBUF - 1;
    auto dma = [&](int st, int u) {
        float* dst = smem + st * STAGE + (wave * NDMA + u) * 256;
        if (ADDR == 0) __builtin_amdgcn_global_load_lds((gptr_t)(sp + u * 256), (lptr_t)dst, 16, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, voff, soff + u * 1024, 0, 0);
    };
    auto adv = [&]() { if (ADDR == 0) sp += 4 * NDMA * 256; else soff += 4 * NDMA * 1024; };
    auto rd = [&](int buf, int g, f32x4 (&af)[TM], f32x4 (&bf)[2]) {
        const int cg = 4 * (2 * g + lh);
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + a_row + i * 32 * 16 + cg);
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f32x4*>(smem + buf * STAGE + b_row + j * 32 * 16 + cg);
    };
    auto mm = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[2]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);
    };
    constexpr int NM = 8 * TM, NR = TM + 2;           // MFMAs / ds_reads per k-group
    if (NDMA) {
        for (int t = 0; t < NBUF - 1; ++t) { for (int u = 0; u < NDMA; ++u) dma(t, u); adv(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    constexpr int D0 = PLACE == 2 ? NDMA / 2 : NDMA, D1 = NDMA - D0;
    for (int kt = 0; kt < steps; ++kt) {
        const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
        rd(cur, 1, a1, b1);
        for (int u = 0; u < D0; ++u) dma(stg, u);
        mm(a0, b0);
        if (PLACE == 1) __builtin_amdgcn_sched_group_barrier(0x010, D0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        if (PLACE != 1) for (int u = 0; u < D0; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NDMA + D0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rd(nxt, 0, a0, b0);
        for (int u = D0; u < NDMA; ++u) dma(stg, u);
        mm(a1, b1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 1);
        for (int u = 0; u < D1; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); }
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 1);
        __builtin_amdgcn_sched_barrier(0);
        adv();
        cur = nxt; stg = stg + 1 == NBUF ? 0 : stg + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int TM, int NBUF, int NDMA, int ADDR, int PLACE, int BAR>
void run(const char* what, int wg_per_cu, const float* src, size_t src_mb) {
    const int blocks = 256 * wg_per_cu, steps = 4096 / (TM / 2);
    const int need = NBUF * (2 * 32 * TM + 128) * 16 * 4;
    int lds = wg_per_cu == 1 ? 150 * 1024 : 80 * 1024;
    if (lds < need) lds = need;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<TM, NBUF, NDMA, ADDR, PLACE, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((loop<TM, NBUF, NDMA, ADDR, PLACE, BAR>), dim3(blocks), dim3(256), lds, 0, src, src_mb * 1024 * 1024 / 4, out, steps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    const double flops = (double)blocks * 4 * steps * (16.0 * TM) * 4096.0;
    printf("%-64s TM %d NBUF %d NDMA %d ADDR %d PLACE %d BAR %d WG/CU %d: %7.2f ms %6.1f TF/s (%.1f %%)%s\n", what, TM, NBUF, NDMA, ADDR, PLACE, BAR, wg_per_cu, best,
           flops / best / 1e9, flops / best / 1e9 / 1.573, e == hipSuccess ? "" : hipGetErrorString(e));
    (void)hipFree(out);
}

int main() {
    float* src; const size_t big = 1024;
    (void)hipMalloc(&src, big * 1024 * 1024);
    (void)hipMemset(src, 0x3c, big * 1024 * 1024);
    run<2, 4, 0, 0, 0, 1>("no operand stream", 1, src, big);
    run<2, 4, 0, 0, 0, 0>("no operand stream, no barrier", 1, src, big);
    run<2, 4, 4, 0, 0, 1>("64x64/wave, global_load_lds, spread (production form)", 1, src, big);
    run<2, 4, 4, 0, 0, 0>("... no barrier", 1, src, big);
    run<2, 4, 4, 1, 0, 1>("buffer_load lds + SGPR offsets, spread", 1, src, big);
    run<2, 4, 4, 0, 1, 1>("global_load_lds, burst at step start", 1, src, big);
    run<2, 4, 4, 1, 1, 1>("buffer_load lds, burst at step start", 1, src, big);
    run<2, 4, 4, 0, 2, 1>("global_load_lds, 2 + 2 over both k-groups", 1, src, big);
    run<2, 4, 4, 1, 2, 1>("buffer_load lds, 2 + 2 over both k-groups", 1, src, big);
    run<2, 8, 4, 1, 2, 1>("buffer_load lds, 2 + 2, 8 stages", 1, src, big);
    run<4, 4, 6, 0, 0, 1>("128x64/wave (256x128 tile), global_load_lds, spread", 1, src, big);
    run<4, 4, 6, 1, 2, 1>("128x64/wave, buffer_load lds, 3 + 3", 1, src, big);
    run<4, 4, 6, 1, 2, 0>("128x64/wave, buffer_load lds, 3 + 3, no barrier", 1, src, big);
    run<2, 4, 4, 1, 2, 1>("64x64/wave, buffer_load lds, 2 + 2", 2, src, big);
    run<2, 4, 4, 1, 0, 1>("64x64/wave, buffer_load lds, spread", 2, src, big);
    return 0;
}
