// Micro-benchmark (GPU box): what each ingredient of the fp32 GEMM K loop costs, on-chip, with nothing else in the way.
//   hipcc --offload-arch=gfx950 -O3 tools/loop_bench.hip -o /tmp/loop_bench && /tmp/loop_bench
// One 256-thread workgroup = 4 waves, each a 64x64 accumulator tile (2x2 MFMA tiles of 32x32), 32 MFMAs per 16-deep K step, exactly the
// production loop's instruction mix.  FLAGS: 1 = fragment reads from LDS (ds_read_b128 x8 per step), 2 = one s_barrier per step,
// 4 = LDS-DMA of the next tile (4 x 1 KiB per wave per step) from a source of `src_mb` MB walked sequentially, counted vmcnt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// NDMA: LDS-DMA pieces per wave per step (4 = a 128x128 tile's 16 KB over 4 waves; 2 ~ a 256x256 tile's share).  FLAGS & 8: stage through
// registers instead (global_load_dwordx4 + ds_write_b128), FLAGS & 16: s_setprio(1) around the MFMA runs.
template <int FLAGS, int NBUF, int NDMA = 4>
__global__ __launch_bounds__(256) void loop(const float* __restrict__ src, size_t src_floats, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];          // NBUF stages x (128 + 128) rows x 16 floats
    constexpr int STAGE = 256 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < NBUF * STAGE; i += 256) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 a0[2], b0[2], a1[2], b1[2];
    for (int i = 0; i < 2; ++i) { a0[i] = f32x4{0.5f + lane * 1e-3f, 0.25f, 0.125f, 1.f}; b0[i] = f32x4{0.3f, 0.7f + lane * 1e-3f, 0.2f, 0.9f}; a1[i] = b0[i]; b1[i] = a0[i]; }
    const int a_row = ((wave >> 1) * 64 + l31) * 16, b_row = (128 + (wave & 1) * 64 + l31) * 16;
    // DMA source: this workgroup walks its own 16 KB-per-step stream through the source buffer
    size_t pos = ((size_t)blockIdx.x * 4099 * 4096) % src_floats;
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
    if (FLAGS & 4) {
        for (int t = 0; t < NBUF - 1; ++t) {
            for (int u = 0; u < 4; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(sp + u * 256), (lptr_t)(smem + t * STAGE + (wave * 4 + u) * 256), 16, 0, 0);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    f32x4 stg_r[4];
    for (int kt = 0; kt < steps; ++kt) {
        const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
        if (FLAGS & 1) rd(cur, 1, a1, b1);
        if ((FLAGS & 4) && !(FLAGS & 8)) {
            for (int u = 0; u < NDMA; ++u) __builtin_amdgcn_global_load_lds((gptr_t)(sp + u * 256), (lptr_t)(smem + stg * STAGE + (wave * 4 + u) * 256), 16, 0, 0);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        if (FLAGS & 8) {           // register staging: write the tile loaded one step ago, request the next
            if (kt) for (int u = 0; u < NDMA; ++u) *reinterpret_cast<f32x4*>(smem + stg * STAGE + (wave * 4 + u) * 256 + lane * 4) = stg_r[u];
            for (int u = 0; u < NDMA; ++u) stg_r[u] = *reinterpret_cast<const f32x4*>(sp + u * 256);
            sp += 4096; pos += 4096; if (pos + 8192 > src_floats) { sp -= pos; pos = 0; }
        }
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(1);
        mm(a0, b0);
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (FLAGS & 1) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        if ((FLAGS & 4) && !(FLAGS & 8)) for (int u = 0; u < NDMA; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
        if (FLAGS & 8) for (int u = 0; u < NDMA; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                                                         __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_barrier(0);
        if ((FLAGS & 4) && !(FLAGS & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NDMA) : "memory");
        if (FLAGS & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        __builtin_amdgcn_sched_barrier(0);
        if (FLAGS & 1) rd(nxt, 0, a0, b0);
        mm(a1, b1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        if (FLAGS & 1) __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 15, 1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt; stg = stg + 1 == NBUF ? 0 : stg + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int FLAGS, int NBUF, int NDMA = 4>
void run(const char* what, int wg_per_cu, int pad_kb, const float* src, size_t src_mb) {
    const int blocks = 256 * wg_per_cu, steps = 4096;
    const int lds = NBUF * 256 * 16 * 4 + pad_kb * 1024;          // padding forces the intended residency (160 KB per CU)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<FLAGS, NBUF, NDMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<FLAGS, NBUF, NDMA>), dim3(blocks), dim3(256), lds, 0, src, src_mb * 1024 * 1024 / 4, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * steps * 32 * 4096.0;
    const double gbs = (FLAGS & 12) ? (double)blocks * steps * 4096.0 * NDMA / best / 1e6 : 0;
    printf("%-58s WG/CU %d stages %d: %7.2f ms %6.1f TF/s (%.0f %% of 157.3)  DMA %.0f GB/s\n", what, wg_per_cu, NBUF, best, flops / best / 1e9, flops / best / 1e9 / 1.573, gbs);
    hipFree(out);
}

int main() {
    float* src; const size_t big = 2048;
    hipMalloc(&src, big * 1024 * 1024);
    hipMemset(src, 0x3c, big * 1024 * 1024);
    run<3, 4>("MFMA + fragment reads + barrier", 2, 0, src, 8);
    run<7, 4, 4>("+ 4 LDS-DMA pieces per wave-step (128x128 tile)", 2, 0, src, big);
    run<7, 4, 3>("+ 3 pieces (256x128 tile's share)", 2, 0, src, big);
    run<7, 4, 2>("+ 2 pieces (256x256 tile's share)", 2, 0, src, big);
    run<7, 4, 1>("+ 1 piece", 2, 0, src, big);
    run<23, 4, 4>("4 pieces, s_setprio(1) around the MFMA runs", 2, 0, src, big);
    run<11, 4, 4>("register staging: 4 x (global_load_dwordx4 + ds_write_b128)", 2, 0, src, big);
    run<11, 4, 2>("register staging: 2 x", 2, 0, src, big);
    run<7, 4, 4>("4 pieces", 1, 80, src, big);
    run<7, 4, 2>("2 pieces", 1, 80, src, big);
    run<11, 4, 4>("register staging 4 x", 1, 80, src, big);
    return 0;
}
