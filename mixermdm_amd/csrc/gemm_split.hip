// fp32 linear layer computed on the bf16 matrix cores by EXACT 3-way operand splitting (gfx950).
//
// Every fp32 value x is the exact sum of three bf16 numbers x1 + x2 + x3 (8 + 8 + 8 significand bits = fp32's 24):
//   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2).
// A product a*w then expands into nine bf16 x bf16 terms, each of which the MFMA forms exactly in fp32; the three terms
// a2*w3, a3*w2, a3*w3 are below 2^-32 of |a||w| and are dropped, the other six are accumulated in fp32:
//   a*w ~= a1w1 + a1w2 + a2w1 + a1w3 + a3w1 + a2w2          (relative truncation error <= 3 * 2^-32 per product)
// which is below the rounding error of one fp32 FMA (2^-24).  Measured against a float64 product the result is as accurate as
// the native v_mfma_f32_32x32x2_f32 kernel (tests/test_gpu_kernels.py::test_linear_split_*) -- and it runs on
// v_mfma_f32_32x32x16_bf16, whose dense rate is 16x the fp32 MFMA's: six of them cost 6/16 of one fp32 MFMA pass.
//
// Operands arrive pre-split as three bf16 planes [3][rows][K] (plane stride given): weights are split once at mmdm_prepare,
// activations by the kernels that produce them (AdaLN, attention, the GELU epilogue here).  Structure = gemm_bf16_kernel with
// three planes per operand tile: LDS-DMA staged, XOR-swizzled 64-byte rows, swapped operands (row on the lane), 16-byte epilogue.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct SArgs {
    const __bf16* A; const __bf16* W; const float* bias; void* C; const float* extra;
    size_t pa, pw, pc;                    // plane strides (elements) of A, W and of a split output
    int lda, ldw, ldc, ld_extra;
    int M, N, K, epilogue, period, out_split;
    int mt, nt, ablate;
    int row0;                             // global index of row 0 (PE epilogue of a row-sliced launch)
    __bf16* P2; size_t p2_plane; int p2_cols, ld2;   // optional second output: columns [0, p2_cols) also as three bf16 planes (attention Q/K operands)
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

template <int TM_, int TN_>
struct SCfg {
    static constexpr int WGM = TM_ / 10, WGN = TN_ / 10, TM = TM_ % 10, TN = TN_ % 10;
    static constexpr int NWAVES = WGM * WGN, THREADS = 64 * NWAVES;
    static constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = 32;          // K step in bf16 elements (64 bytes)
    static constexpr int A_PLANE = BM * 16, B_PLANE = BN * 16;                    // one plane of a tile, 4-byte units
    static constexpr int A_FLOATS = 3 * A_PLANE, B_FLOATS = 3 * B_PLANE;
    static constexpr int SMEM_BYTES = 2 * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int NAP = BM / 16, NBP = BN / 16;                            // 1-KiB pieces per plane
    static constexpr int NA = 3 * NAP, NB = 3 * NBP;
    static constexpr int NI = (NA + NB) / NWAVES;
    static_assert((NA + NB) % NWAVES == 0, "pieces must divide evenly over the waves");
};

template <int TM_, int TN_>
__global__ __launch_bounds__((SCfg<TM_, TN_>::THREADS)) void gemm_split_kernel(SArgs p) {
    using C_ = SCfg<TM_, TN_>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [2][3][BM*16]
    float* Bs = smem + 2 * C_::A_FLOATS;           // [2][3][BN*16]

    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (swz / p.nt) * BM;
    const int n0 = (swz % p.nt) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C_::WGN, wn = wave % C_::WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    const char* src[C_::NI];
    int dst[C_::NI];
    bool isa[C_::NI];
#pragma unroll
    for (int u = 0; u < C_::NI; ++u) {
        const int pq = wave + C_::NWAVES * u;
        const int prow = lane >> 2, pc = lane & 3;
        isa[u] = pq < C_::NA;
        const int pl = isa[u] ? pq / C_::NAP : (pq - C_::NA) / C_::NBP;          // plane
        const int pp = isa[u] ? pq % C_::NAP : (pq - C_::NA) % C_::NBP;          // piece inside the plane
        const int trow = 16 * pp + prow;
        const int gch = pc ^ ((trow >> 2) & 3);
        if (isa[u]) {
            int grow = m0 + trow;
            grow = grow < p.M ? grow : p.M - 1;
            src[u] = reinterpret_cast<const char*>(p.A + (size_t)pl * p.pa + (size_t)grow * p.lda) + 16 * gch;
            dst[u] = pl * C_::A_PLANE + 16 * pp * 16;
        } else {
            int grow = n0 + trow;
            grow = grow < p.N ? grow : p.N - 1;
            src[u] = reinterpret_cast<const char*>(p.W + (size_t)pl * p.pw + (size_t)grow * p.ldw) + 16 * gch;
            dst[u] = 2 * C_::A_FLOATS + pl * C_::B_PLANE + 16 * pp * 16;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < C_::NI; ++u) {
            const int boff = isa[u] ? buf * C_::A_FLOATS : buf * C_::B_FLOATS;
            __builtin_amdgcn_global_load_lds((gptr_t)src[u], (lptr_t)(smem + dst[u] + boff), 16, 0, 0);
            src[u] += 64;
        }
    };

    // Accumulators start as bias (+ residual / + PE row).  All loads are issued back to back behind WAVE-UNIFORM branches and waited for
    // once: with per-element runtime conditions hipcc branches around every load and waits vmcnt(0) after each one -- 16-32 serialized
    // L2 round trips per tile, fully exposed at one workgroup per CU.  Rows / columns past the edge read a clamped (valid) address: their
    // accumulators are never stored.
    f32x16 acc[TM][TN];
    {
        const bool has_bias = p.bias != nullptr;
        const bool has_ext = p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE;
        f32x4 bv[TN][4], ev[TM][TN][4];
        int colc[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                colc[j][qd] = min(n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh, p.N - 4);
                bv[j][qd] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (has_bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) bv[j][qd] = *reinterpret_cast<const f32x4*>(p.bias + colc[j][qd]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) ev[i][j][qd] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_ext) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rowc = min(m0 + wm * (32 * TM) + i * 32 + l31, p.M - 1);
                const int er = p.epilogue == MMDM_EPI_BIAS_PE ? (rowc + p.row0) % p.period : rowc;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) ev[i][j][qd] = *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + colc[j][qd]);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 v = bv[j][qd] + ev[i][j][qd];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] = v[c];
                }
    }

    const int nkt = p.K / BK;
    stage(0);

    const int sw = (l31 >> 2) & 3;
    const int a_row = (wm * (32 * TM) + l31) * 16;
    const int b_row = (wn * (32 * TN) + l31) * 16;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt && !(p.ablate & 1)) stage(cur ^ 1);
        const float* Ac = As + (p.ablate & 1 ? 0 : cur) * C_::A_FLOATS + a_row;
        const float* Bc = Bs + (p.ablate & 1 ? 0 : cur) * C_::B_FLOATS + b_row;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int cg = 4 * ((2 * kb + lh) ^ sw);
            bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Ac + pl * C_::A_PLANE + i * 32 * 16 + cg));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bc + pl * C_::B_PLANE + j * 32 * 16 + cg));
            }
            // six product terms, smallest first; each pass walks all TM x TN accumulators so dependent MFMAs are TM*TN apart
            constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[PB[t]][j], af[PA[t]][i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]));

#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * (32 * TM) + i * 32 + l31;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int col = n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh;
                if (col >= p.N) continue;
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t = acc[i][j][4 * qd + c];
                    if (p.epilogue == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                    else if (p.epilogue == MMDM_EPI_BIAS_SILU) t = silu(t);
                    v[c] = t;
                }
                const bool second = p.P2 && col < p.p2_cols;
                if (p.out_split || second) {
                    bf16x4 o1, o2, o3;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        o1[c] = (__bf16)v[c];
                        const float r1 = v[c] - (float)o1[c];
                        o2[c] = (__bf16)r1;
                        o3[c] = (__bf16)(r1 - (float)o2[c]);
                    }
                    if (p.out_split) {
                        __bf16* cp = static_cast<__bf16*>(p.C) + (size_t)row * p.ldc + col;
                        *reinterpret_cast<bf16x4*>(cp) = o1;
                        *reinterpret_cast<bf16x4*>(cp + p.pc) = o2;
                        *reinterpret_cast<bf16x4*>(cp + 2 * p.pc) = o3;
                    }
                    if (second) {
                        __bf16* cp = p.P2 + (size_t)(row + p.row0) * p.ld2 + col;
                        *reinterpret_cast<bf16x4*>(cp) = o1;
                        *reinterpret_cast<bf16x4*>(cp + p.p2_plane) = o2;
                        *reinterpret_cast<bf16x4*>(cp + 2 * p.p2_plane) = o3;
                    }
                }
                if (!p.out_split) *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + (size_t)row * p.ldc + col) = v;
            }
    }
}

template <int TM_, int TN_>
int launch(SArgs a, hipStream_t st) {
    using C_ = SCfg<TM_, TN_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    mmdm_note_gemm("gemm_split<%d,%d>", TM_, TN_);
    hipLaunchKernelGGL((gemm_split_kernel<TM_, TN_>), dim3(a.mt * a.nt), dim3(C_::THREADS), C_::SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_split");
}

template <int TM_, int TN_>
int set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<TM_, TN_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SCfg<TM_, TN_>::SMEM_BYTES);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_split): %s", hipGetErrorString(e));
    return MMDM_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------
// Deep-pipelined variant: K step 16 (32-byte LDS rows), NS-stage ring, counted vmcnt.
//
// Why: at the bf16 MFMA rate a 256x128 tile consumes 72 KB of operand planes per 0.64 us, and ~17 % of those lines miss the
// 4 MiB XCD L2 (panels of K = 1024 are MBs) and come from the Infinity Cache 1-2 us later: with one stage of prefetch the K step
// lasts as long as the slowest miss (measured 6.4 TB/s of L2->LDS traffic, 37 % of the split roof).  This kernel keeps NS-1
// stages in flight (`s_waitcnt vmcnt(loads of the newer stages)`, never 0 inside the loop) and moves fewer bytes per flop
// (256x256 tile: 48 KB per stage of 16 K-elements).  32-byte rows make the LDS image of a 32-row fragment 1 KiB contiguous: the
// DMA piece and the ds_read_b128 fragment read are conflict-free without a swizzle.
template <int TM_, int TN_, int NS_>
struct PCfg {
    static constexpr int WGM = TM_ / 10, WGN = TN_ / 10, TM = TM_ % 10, TN = TN_ % 10, NS = NS_;
    static constexpr int NWAVES = WGM * WGN, THREADS = 64 * NWAVES;
    static constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = 16;
    static constexpr int A_PLANE = BM * 8, B_PLANE = BN * 8;                      // 4-byte units (32 B per row)
    static constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    static constexpr int SMEM_BYTES = NS * STAGE * 4;
    static constexpr int NAP = BM / 32, NBP = BN / 32;                            // 1-KiB pieces (32 rows x 32 B) per plane
    static constexpr int NA = 3 * NAP, NB = 3 * NBP;
    // DMA instructions per wave per stage; when the pieces do not divide evenly the surplus slots re-load the first pieces
    // (same source, same destination: harmless) so that every wave counts the same number of loads per stage
    static constexpr int NI = (NA + NB + NWAVES - 1) / NWAVES;
    static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
};

// LDS-DMA issued from inline asm: hipcc (ROCm 7.2) otherwise treats every pending LDS-DMA as a possible writer of whatever a later
// ds_read reads and drains the whole queue (s_waitcnt vmcnt(0)) before the first fragment is used -- which is exactly the overlap a
// multi-stage ring exists for.  The queue is counted by hand instead (wait_vmcnt below).  LDS address = m0 + lane * 16.
__device__ __forceinline__ void glds16(const void* gsrc, const float* lds_dst) {
    const unsigned lds = (unsigned)(size_t)(lptr_t)lds_dst;
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds));      // no "memory" clobber: see above
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <class C_>
__device__ __forceinline__ void pipe_step(float* __restrict__ wr, const float* __restrict__ rd, bool do_stage, const char* (&src)[C_::NI], const int (&dst)[C_::NI],
                                          f32x16 (&acc)[C_::TM][C_::TN], int a_row, int b_row, bool do_read, bf16x8 (&af)[3][C_::TM], bf16x8 (&bf)[3][C_::TN]) {
    constexpr int TM = C_::TM, TN = C_::TN, NI = C_::NI;
    if (do_read) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(rd + b_row + pl * C_::B_PLANE + j * 32 * 8));
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(rd + a_row + pl * C_::A_PLANE + i * 32 * 8));
        }
    }
    // The stage's DMA instructions are issued ONE PER GROUP OF MFMAs, not in a burst behind the barrier: an LDS-DMA issue holds the wave's
    // instruction stream for ~20 cycles (per-lane addresses), and with both waves of a SIMD at the same program point a burst of NI of
    // them leaves the MFMA pipe empty; between MFMAs the issue hides under the 32-cycle MFMA in flight.  sched_barrier pins the order.
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
    constexpr int PER = (NI + 5) / 6;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[PB[t]][j], af[PA[t]][i], acc[i][j], 0, 0, 0);
        if (do_stage) {
#pragma unroll
            for (int u = t * PER; u < (t + 1) * PER && u < NI; ++u) {
                glds16(src[u], wr + dst[u]);
                src[u] += 2 * C_::BK;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int TM_, int TN_, int NS_>
__global__ __launch_bounds__((PCfg<TM_, TN_, NS_>::THREADS)) void gemm_split_pipe_kernel(SArgs p) {
    using C_ = PCfg<TM_, TN_, NS_>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN, NS = C_::NS, NI = C_::NI;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [NS][A planes 3*BM*8 | B planes 3*BN*8]

    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // tiles that run together on an XCD form GM x nt groups walked n-fastest: a W panel is shared by GM concurrent tiles
    constexpr int GM = 4;
    const int grp = swz / (GM * p.nt), in = swz % (GM * p.nt);
    const int gm = min(GM, p.mt - grp * GM);
    const int m0 = (grp * GM + in % gm) * BM;
    const int n0 = (in / gm) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C_::WGN, wn = wave % C_::WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    const char* src[NI];
    int dst[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        int pq = wave + C_::NWAVES * u;
        if (pq >= C_::NA + C_::NB) pq -= C_::NA + C_::NB;
        const int prow = lane >> 1, pc = lane & 1;
        const bool isa = pq < C_::NA;
        const int pl = isa ? pq / C_::NAP : (pq - C_::NA) / C_::NBP;
        const int pp = isa ? pq % C_::NAP : (pq - C_::NA) % C_::NBP;
        const int trow = 32 * pp + prow;
        if (isa) {
            int grow = m0 + trow;
            grow = grow < p.M ? grow : p.M - 1;
            src[u] = reinterpret_cast<const char*>(p.A + (size_t)pl * p.pa + (size_t)grow * p.lda) + 16 * pc;
            dst[u] = pl * C_::A_PLANE + 32 * pp * 8;
        } else {
            int grow = n0 + trow;
            grow = grow < p.N ? grow : p.N - 1;
            src[u] = reinterpret_cast<const char*>(p.W + (size_t)pl * p.pw + (size_t)grow * p.ldw) + 16 * pc;
            dst[u] = 3 * C_::A_PLANE + pl * C_::B_PLANE + 32 * pp * 8;
        }
    }
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            glds16(src[u], smem + buf * C_::STAGE + dst[u]);
            src[u] += 2 * BK;
        }
    };

    // Accumulators start as bias (+ residual / + PE row).  All loads are issued back to back behind WAVE-UNIFORM branches and waited for
    // once: with per-element runtime conditions hipcc branches around every load and waits vmcnt(0) after each one -- 16-32 serialized
    // L2 round trips per tile, fully exposed at one workgroup per CU.  Rows / columns past the edge read a clamped (valid) address: their
    // accumulators are never stored.
    f32x16 acc[TM][TN];
    {
        const bool has_bias = p.bias != nullptr;
        const bool has_ext = p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE;
        f32x4 bv[TN][4], ev[TM][TN][4];
        int colc[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                colc[j][qd] = min(n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh, p.N - 4);
                bv[j][qd] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (has_bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) bv[j][qd] = *reinterpret_cast<const f32x4*>(p.bias + colc[j][qd]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) ev[i][j][qd] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_ext) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rowc = min(m0 + wm * (32 * TM) + i * 32 + l31, p.M - 1);
                const int er = p.epilogue == MMDM_EPI_BIAS_PE ? (rowc + p.row0) % p.period : rowc;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) ev[i][j][qd] = *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + colc[j][qd]);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 v = bv[j][qd] + ev[i][j][qd];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] = v[c];
                }
    }

    const int nkt = p.K / BK;
    // The bias / residual loads that initialise the accumulators must be retired HERE: left pending, the compiler waits for them at
    // their first use -- the first MFMA of the loop body -- with an s_waitcnt vmcnt(0) that also drains the DMA ring every iteration.
    // An opaque asm that touches each accumulator makes that wait land before the loop.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // prologue: NS-1 stages in flight
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nkt) stage(s);

    const int a_row = (wm * (32 * TM) + l31) * 8 + 4 * lh;
    const int b_row = 3 * C_::A_PLANE + (wn * (32 * TN) + l31) * 8 + 4 * lh;

    bf16x8 af[3][TM], bf[3][TN];
    int buf = 0, nbuf = NS - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed when at most the loads of the (NS-2) newer stages are still in flight
        if (kt + NS - 1 <= nkt) wait_vmcnt<(NS - 2) * NI>();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tail: fewer stages behind it
        __builtin_amdgcn_s_barrier();
        // DMA destination and fragment source are different ring slots: passed as __restrict__ arguments of one inlined function so
        // that the compiler's LDS-DMA tracking (alias scopes) does not drain the DMA queue (vmcnt(0)) before the fragment reads are used
        pipe_step<C_>(smem + nbuf * C_::STAGE, smem + buf * C_::STAGE, kt + NS - 1 < nkt && !(p.ablate & 1), src, dst, acc, a_row, b_row,
                      !(p.ablate & 2) || kt == 0, af, bf);
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]));

#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * (32 * TM) + i * 32 + l31;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int col = n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh;
                if (col >= p.N) continue;
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float t = acc[i][j][4 * qd + c];
                    if (p.epilogue == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                    else if (p.epilogue == MMDM_EPI_BIAS_SILU) t = silu(t);
                    v[c] = t;
                }
                const bool second = p.P2 && col < p.p2_cols;
                if (p.out_split || second) {
                    bf16x4 o1, o2, o3;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        o1[c] = (__bf16)v[c];
                        const float r1 = v[c] - (float)o1[c];
                        o2[c] = (__bf16)r1;
                        o3[c] = (__bf16)(r1 - (float)o2[c]);
                    }
                    if (p.out_split) {
                        __bf16* cp = static_cast<__bf16*>(p.C) + (size_t)row * p.ldc + col;
                        *reinterpret_cast<bf16x4*>(cp) = o1;
                        *reinterpret_cast<bf16x4*>(cp + p.pc) = o2;
                        *reinterpret_cast<bf16x4*>(cp + 2 * p.pc) = o3;
                    }
                    if (second) {
                        __bf16* cp = p.P2 + (size_t)(row + p.row0) * p.ld2 + col;
                        *reinterpret_cast<bf16x4*>(cp) = o1;
                        *reinterpret_cast<bf16x4*>(cp + p.p2_plane) = o2;
                        *reinterpret_cast<bf16x4*>(cp + 2 * p.p2_plane) = o3;
                    }
                }
                if (!p.out_split) *reinterpret_cast<f32x4*>(static_cast<float*>(p.C) + (size_t)row * p.ldc + col) = v;
            }
    }
}

template <int TM_, int TN_, int NS_>
int launch_pipe(SArgs a, hipStream_t st) {
    using C_ = PCfg<TM_, TN_, NS_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    mmdm_note_gemm("gemm_split_pipe<%d,%d,%d>", TM_, TN_, NS_);
    hipLaunchKernelGGL((gemm_split_pipe_kernel<TM_, TN_, NS_>), dim3(a.mt * a.nt), dim3(C_::THREADS), C_::SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_split_pipe");
}

template <int TM_, int TN_, int NS_>
int set_attr_pipe() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_pipe_kernel<TM_, TN_, NS_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PCfg<TM_, TN_, NS_>::SMEM_BYTES);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_split_pipe): %s", hipGetErrorString(e));
    return MMDM_OK;
}

// x -> three bf16 planes out[0], out[plane], out[2*plane]; exact: x == out0 + out1 + out2 in real arithmetic
__global__ void split3_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n, size_t plane) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = in[i];
        const __bf16 b1 = (__bf16)x;
        const float r1 = x - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        out[i] = b1;
        out[plane + i] = b2;
        out[2 * plane + i] = (__bf16)(r1 - (float)b2);
    }
}

int g_split_cfg = -1;
int g_split_ablate = 0;

}  // namespace

int mmdm_gemm_split_init(void) {
    int rc;
    if ((rc = set_attr<22, 22>())) return rc;
    if ((rc = set_attr<42, 22>())) return rc;
    if ((rc = set_attr<22, 21>())) return rc;
    if ((rc = set_attr<24, 22>())) return rc;
    if ((rc = set_attr_pipe<24, 42, 3>())) return rc;     // 256 x 256, 8 waves (2 x 4), 128 x 64 per wave
    if ((rc = set_attr_pipe<24, 22, 4>())) return rc;     // 256 x 128, 4 waves (2 x 2), 128 x 64 per wave
    if ((rc = set_attr_pipe<22, 22, 6>())) return rc;     // 128 x 128, 4 waves
    if ((rc = set_attr_pipe<42, 22, 4>())) return rc;     // 256 x 128, 8 waves (4 x 2), 64 x 64 per wave
    const char* e = getenv("MMDM_SPLIT_CFG");
    g_split_cfg = e ? atoi(e) : -1;
    return MMDM_OK;
}

extern "C" void mmdmx_set_split_cfg(int c) { g_split_cfg = c; }
extern "C" void mmdmx_set_split_ablate(int c) { g_split_ablate = c; }

extern "C" int mmdm_f32_split3(const float* in, void* out, int64_t n, int64_t plane_stride, void* stream) {
    if (n <= 0) return MMDM_OK;
    if (!in || !out || plane_stride < n) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_f32_split3: bad arguments");
    hipLaunchKernelGGL(split3_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), in, static_cast<__bf16*>(out), (size_t)n, (size_t)plane_stride);
    return mmdm_check_launch("f32_split3");
}

extern "C" int mmdm_linear_split(const void* A, int lda, int64_t a_plane, const void* W, int ldw, int64_t w_plane, const float* bias, void* C, int ldc,
                                 int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_split_ex(A, lda, a_plane, W, ldw, w_plane, bias, C, ldc, c_plane, out_split, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, 0, stream);
}

int mmdm_linear_split_ex(const void* A, int lda, int64_t a_plane, const void* W, int ldw, int64_t w_plane, const float* bias, void* C, int ldc,
                         int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period,
                         void* planes2, int ld2, int64_t plane2_stride, int planes2_cols, void* stream) {
    mmdm_note_gemm_reset();
    if (M == 0 || N == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    if (!A || !W || !C || M < 0 || N < 0 || K <= 0 || lda < K || ldw < K || ldc < N || a_plane <= 0 || w_plane <= 0 || (out_split && c_plane <= 0))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_split: bad shape M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
    if (epilogue < MMDM_EPI_BIAS || epilogue > MMDM_EPI_BIAS_SILU) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_split: unknown epilogue %d", epilogue);
    const bool ext = epilogue == MMDM_EPI_BIAS_RESID || epilogue == MMDM_EPI_BIAS_PE;
    if (ext && (!extra || ld_extra < N)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_split: epilogue %d needs `extra` with ld >= N", epilogue);
    if (epilogue == MMDM_EPI_BIAS_PE && period <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_split: PE epilogue needs period > 0");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((K & 31) || (lda & 7) || (ldw & 7) || (a_plane & 7) || (w_plane & 7) || !al16(A) || !al16(W))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split: needs K %% 32 == 0 and 16-byte aligned bf16 rows / planes");
    if ((N & 3) || (ldc & 3) || (c_plane & 3) || !al16(C) || (bias && !al16(bias)) || (ext && ((ld_extra & 3) || !al16(extra))))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split: needs N %% 4 == 0 and 16-byte aligned output / bias / residual rows");
    SArgs a;
    a.A = static_cast<const __bf16*>(A); a.W = static_cast<const __bf16*>(W); a.bias = bias; a.C = C; a.extra = extra;
    a.pa = (size_t)a_plane; a.pw = (size_t)w_plane; a.pc = (size_t)c_plane;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ld_extra = ld_extra;
    a.M = M; a.N = N; a.K = K; a.epilogue = epilogue; a.period = period > 0 ? period : 1; a.out_split = out_split;
    a.mt = a.nt = 0; a.ablate = g_split_ablate; a.row0 = 0;
    a.P2 = static_cast<__bf16*>(planes2); a.p2_plane = (size_t)plane2_stride; a.p2_cols = planes2_cols; a.ld2 = ld2;
    if (planes2 && ((ld2 & 3) || (plane2_stride & 3) || (planes2_cols & 3) || !al16(planes2)))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split: second output needs 8-byte aligned bf16 rows / planes");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (g_split_cfg) {
        case 0: return launch<22, 22>(a, st);
        case 1: return launch<42, 22>(a, st);
        case 2: return launch<22, 21>(a, st);
        case 4: return launch<24, 22>(a, st);
        case 10: return launch_pipe<24, 42, 3>(a, st);
        case 11: return launch_pipe<24, 22, 4>(a, st);
        case 12: return launch_pipe<22, 22, 6>(a, st);
        case 13: return launch_pipe<42, 22, 4>(a, st);
        default: break;
    }
    // measured on M = 19 200 (tools/gemm_split_bench.py): 128x64 tiles for the N = 512 mixer GEMMs (more tiles than CUs), 256x128 otherwise
    if (N <= 512) return launch<22, 21>(a, st);
    // 256x128 tiles run one workgroup per CU (144 KB of LDS), so a launch takes ceil(tiles / 256) rounds of equal-length tiles and a
    // partly filled last round is pure loss (N = 1024, M = 19 200: 600 tiles = 2.34 -> 3 rounds).  Hybrid tiling: the leading M-tiles
    // that fill whole rounds go to the 256x128 kernel, the remaining rows to the 128x64 kernel (2 workgroups per CU, 4x finer tiles).
    int ncu = 256;
    const int mt = (M + 255) / 256, nt = (N + 127) / 128;
    int m_main = mt;
    // measured (tools/gemm_split_bench.py, M = 19 200): +5 % at N = 1024 (K = 1024 and 2048); at N >= 1536 the single launch is faster
    // (tiles are not equally long there: L2 reuse differs along N), so the split is applied to N <= 1024 only
    if ((long)mt * nt > ncu && N <= 1024 && !(g_split_ablate & 4)) {
        for (int m = mt; m >= 1; --m) {
            const long tiles = (long)m * nt, rounds = (tiles + ncu - 1) / ncu;
            if ((double)tiles / (double)(rounds * ncu) >= 0.97) { m_main = m; break; }
        }
    }
    if (m_main >= mt || (long)m_main * 256 >= M) return launch<42, 22>(a, st);
    SArgs b = a;
    const int M1 = m_main * 256;
    a.M = M1;
    if (int rc = launch<42, 22>(a, st)) return rc;
    b.M = M - M1;
    b.row0 = M1;
    b.A = a.A + (size_t)M1 * lda;
    b.C = out_split ? static_cast<void*>(static_cast<__bf16*>(C) + (size_t)M1 * ldc) : static_cast<void*>(static_cast<float*>(C) + (size_t)M1 * ldc);
    if (ext && epilogue == MMDM_EPI_BIAS_RESID) b.extra = extra + (size_t)M1 * ld_extra;
    return launch<22, 21>(b, st);
}
