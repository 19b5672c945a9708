// Micro-benchmark 4 (GPU box): how long does an EPILOGUE-like instruction stream take beside a co-resident workgroup that is in its K loop?
//   hipcc --offload-arch=gfx950 -O3 tools/loop_bench4.hip -o /tmp/loop_bench4 && /tmp/loop_bench4
// Workgroups 0..255 (one per CU, dispatched first) run the fp32 MFMA K loop (buffer-addressed LDS-DMA, 4 stages) for `steps` steps when
// BUSY = 1, or exit at once when BUSY = 0.  Workgroups 256..511 land beside them and run `reps` repetitions of one epilogue form:
//   EPI 0: 1024 independent-ish VALU FMAs per lane        EPI 1: 16 x global_store_dwordx4, row per lane (32 rows x 32 B per instruction)
//   EPI 2: 64 x global_store_dword, column per lane (2 rows x 128 B per instruction)   EPI 3: EPI 1's stores, 64-float accumulators read from AGPR-like state
// and report the mean time per repetition (s_memrealtime, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI, int BUSY, int PRIO>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, float* cbuf, int ldc, int steps, int reps, unsigned long long* times) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NBUF = 4, NDMA = 4, STAGE = 256 * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
    if (blockIdx.x < 256) {
        if (!BUSY) return;
        for (int i = tid; i < NBUF * STAGE; i += blockDim.x) smem[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
        __syncthreads();
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        f32x4 a0[2], b0[2], a1[2], b1[2];
        for (int i = 0; i < 2; ++i) { a0[i] = f32x4{0.5f + lane * 1e-3f, 0.25f, 0.125f, 1.f}; a1[i] = f32x4{0.3f, 0.7f + lane * 1e-3f, 0.2f, 0.9f}; b0[i] = a1[i]; b1[i] = a0[i]; }
        const int a_row = ((wave >> 1) * 64 + l31) * 16, b_row = (128 + (wave & 1) * 64 + l31) * 16;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 1048576), 0, 0x7fffffff, 0x00020000);
        const int voff = (wave * NDMA * 256 + lane * 4) * 4;
        int soff = 0, cur = 0, stg = NBUF - 1;
        auto dma = [&](int st, int u) { __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + st * STAGE + (wave * NDMA + u) * 256), 16, voff, soff + u * 1024, 0, 0); };
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
        for (int t = 0; t < NBUF - 1; ++t) { for (int u = 0; u < NDMA; ++u) dma(t, u); soff += 4 * NDMA * 1024; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < steps; ++kt) {
            const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
            rd(cur, 1, a1, b1);
            dma(stg, 0); dma(stg, 1);
            mm(a0, b0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            for (int u = 0; u < 2; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 3) * NDMA + 2) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rd(nxt, 0, a0, b0);
            dma(stg, 2); dma(stg, 3);
            mm(a1, b1);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 1);
            for (int u = 0; u < 2; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 1);
            __builtin_amdgcn_sched_barrier(0);
            soff += 4 * NDMA * 1024; if (soff > 3 * 1048576) soff = 0;
            cur = nxt; stg = stg + 1 == NBUF ? 0 : stg + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[(size_t)blockIdx.x * 256 + tid] = s;
        return;
    }
    // ---- the epilogue-like workgroup ----
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const int tile = blockIdx.x - 256;
    float* C = cbuf + (size_t)tile * 128 * ldc;              // a 128 x 128 output tile of a [256*128, ldc] matrix
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 v[16];
    for (int q = 0; q < 16; ++q) v[q] = f32x4{1.f + lane + q, 2.f, 3.f + q, 4.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if (EPI == 0) {
#pragma unroll
            for (int it = 0; it < 16; ++it)
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = v[q] * 1.0001f + v[(q + 1) & 15];
        } else if (EPI == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int row = wm * 64 + i * 32 + l31, col = wn * 64 + j * 32 + 8 * qd + 4 * lh;
                        *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = v[(i * 2 + j) * 4 + qd];
                    }
        } else if (EPI == 3) {
            // the accumulators of a finished tile live in AGPRs: 64 v_accvgpr_read per lane, then the 16 row-per-lane stores
            float t[64];
#pragma unroll
            for (int q = 0; q < 64; ++q) asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(t[q]) : "n"(q));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int row = wm * 64 + i * 32 + l31, col = wn * 64 + j * 32 + 8 * qd + 4 * lh;
                        const int b = ((i * 2 + j) * 4 + qd) * 4;
                        *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = f32x4{t[b], t[b + 1], t[b + 2], t[b + 3]};
                    }
        } else if (EPI == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, col = wn * 64 + j * 32 + l31;
                        C[(size_t)row * ldc + col] = v[(i * 2 + j) * 4 + (e >> 2)][e & 3];
                    }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    float s = 0; for (int q = 0; q < 16; ++q) s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { times[2 * tile] = t1 - t0; times[2 * tile + 1] = t2 - t0; }
}

template <int EPI, int BUSY, int PRIO>
void run(const char* what, const float* src, float* cbuf, int reps) {
    const int lds = 80 * 1024, steps = 3000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<EPI, BUSY, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float* out; (void)hipMalloc(&out, (size_t)512 * 256 * 4);
    unsigned long long* times; (void)hipMalloc(&times, 512 * 8); (void)hipMemset(times, 0, 512 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<EPI, BUSY, PRIO>), dim3(512), dim3(256), lds, 0, src, out, cbuf, 1024, steps, reps, times);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h[512]; (void)hipMemcpy(h, times, 512 * 8, hipMemcpyDeviceToHost);
    double a = 0, b = 0; for (int i = 0; i < 256; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    hipError_t e = hipGetLastError();
    printf("%-60s busy %d prio %d: issue %7.2f us per repetition, with the final drain %7.2f us  %s\n", what, BUSY, PRIO, a / 256 / 100.0 / reps, b / 256 / 100.0 / reps, e == hipSuccess ? "" : hipGetErrorString(e));
    (void)hipFree(out); (void)hipFree(times);
}

int main() {
    float *src, *cbuf;
    (void)hipMalloc(&src, (size_t)256 * 4 * 1048576 + (64 << 20));
    (void)hipMemset(src, 0x3c, (size_t)256 * 4 * 1048576 + (64 << 20));
    (void)hipMalloc(&cbuf, (size_t)256 * 128 * 1024 * 4);
    run<0, 0, 0>("1024 VALU FMAs per lane", src, cbuf, 200);
    run<0, 1, 0>("1024 VALU FMAs per lane", src, cbuf, 200);
    run<0, 1, 1>("1024 VALU FMAs per lane", src, cbuf, 200);
    run<1, 0, 0>("16 x store_dwordx4, row per lane (128x128 tile)", src, cbuf, 200);
    run<1, 1, 0>("16 x store_dwordx4, row per lane", src, cbuf, 200);
    run<1, 1, 1>("16 x store_dwordx4, row per lane", src, cbuf, 200);
    run<3, 1, 0>("64 x v_accvgpr_read + 16 x store_dwordx4 row per lane", src, cbuf, 200);
    run<3, 1, 1>("64 x v_accvgpr_read + 16 x store_dwordx4 row per lane", src, cbuf, 200);
    run<2, 0, 0>("64 x store_dword, column per lane", src, cbuf, 200);
    run<2, 1, 0>("64 x store_dword, column per lane", src, cbuf, 200);
    return 0;
}
