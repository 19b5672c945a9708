// fp32 linear layer computed on the 16-bit matrix cores from two-way fp16 operand splits (gfx950).
//
// Every fp32 operand is carried as two fp16 planes (kernels.h, mmdm_split2):  x ~= h + l / 2048,  h = fp16(x),  l = fp16((x - h) * 2048)
// -- 11 + 11 significand bits, relative representation error <= 2^-22.  A product a*w expands into four fp16 x fp16 terms, each of which the
// MFMA forms exactly in fp32; al*wl is below 2^-22 |a||w| and is dropped, the other three are accumulated in fp32 in TWO accumulators:
//   hi += ah*wh,      lo += ah*wl + al*wh,      result = hi + lo / 2048        (one FMA in the epilogue, then the bias / residual order below)
// The representation error is independent per element and averages over K; the fp32 accumulation chain of ANY fp32 GEMM over K = 1024 terms is
// ten times larger (tools/split_numerics.py: against a float64 product this form is as accurate as the native v_mfma_f32_32x32x2_f32 kernel and
// as the six-term bf16 three-way split of rounds 1-3, which it replaces at half the MFMA work and 4 instead of 6 operand bytes per element:
// tests/test_gpu_kernels.py::test_linear_split_*, tests/test_gpu_headline.py).  Three v_mfma_f32_32x32x16_f16 cost 3/16 of one fp32 MFMA pass.
//
// Operands arrive pre-split as two fp16 planes [2][rows][K] (plane stride given): weights are split once at mmdm_prepare,
// activations by the kernels that produce them (AdaLN, attention, the GELU epilogue here).  Structure = gemm_bf16_kernel with
// two planes per operand tile: LDS-DMA staged, XOR-swizzled 64-byte rows, swapped operands (row on the lane), 16-byte epilogue.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
constexpr int NPL = MMDM_SPLIT_NPL;       // operand planes (kernels.h)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct SArgs {
    const _Float16* A; const _Float16* W; const float* bias; void* C; const float* extra;
    size_t pa, pw, pc;                    // plane strides (elements) of A, W and of a split output
    int lda, ldw, ldc, ld_extra;
    int M, N, K, epilogue, period, out_split;
    int mt, nt, ablate;
    int row0;                             // global index of row 0 (PE epilogue of a row-sliced launch)
    unsigned long long* tl;               // diagnostic: per-wave phase cycle sums (tools/split_timeline.py), null in production
    __bf16* P2; size_t p2_plane; int p2_cols, ld2;   // optional second output: columns [0, p2_cols) also as three exact bf16 planes (Q/K operands of attn_qkp_kernel<DH, 3>)
    int tst;                              // packed-W kernel: results leave through the workgroup's LDS transposition (split_finish_t)
};


template <int TM_, int TN_>
struct SCfg {
    static constexpr int WGM = TM_ / 10, WGN = TN_ / 10, TM = TM_ % 10, TN = TN_ % 10;
    static constexpr int NWAVES = WGM * WGN, THREADS = 64 * NWAVES;
    static constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = 32;          // K step in fp16 elements (64 bytes)
    static constexpr int A_PLANE = BM * 16, B_PLANE = BN * 16;                    // one plane of a tile, 4-byte units
    static constexpr int A_FLOATS = NPL * A_PLANE, B_FLOATS = NPL * B_PLANE;
    static constexpr int SMEM_BYTES = 2 * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int NAP = BM / 16, NBP = BN / 16;                            // 1-KiB pieces per plane
    static constexpr int NA = NPL * NAP, NB = NPL * NBP;
    static constexpr int NI = (NA + NB) / NWAVES;
    static_assert((NA + NB) % NWAVES == 0, "pieces must divide evenly over the waves");
};

// Epilogue shared by the fp32-split kernels: activation, output form (fp32 rows / the two fp16 operand planes) and the optional second plane
// output are chosen ONCE per tile, outside the element loops (with the runtime tests inside them every element carried every variant:
// tens of KB of branchy code that a wave crawls through while a co-resident workgroup owns the matrix pipe; see gemm_f32.hip and
// tools/gemm_timeline.py).  fp32 rows are stored through a buffer resource over the tile's rows: rows past M are dropped by the hardware.
#if defined(__HIP_DEVICE_COMPILE__)
template <int TM, int TN, int BM>
__device__ __forceinline__ void split_finish(const SArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int l31, int lh) {
    const int rows_here = min(p.M - m0, BM);
    auto finish = [&](auto act_c, auto split_c, auto sec_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool SPLIT = decltype(split_c)::value, SEC = decltype(sec_c)::value;
        const int col0 = n0 + wn * (32 * TN) + 4 * lh;
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(p.C) + (size_t)m0 * p.ldc, 0, rows_here * p.ldc * 4, 0x00020000);
        const int voffC = (wm * (32 * TM) + l31) * p.ldc * 4 + col0 * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * (32 * TM) + i * 32 + l31;
            const bool rok = row < p.M;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int col = col0 + j * 32 + 8 * qd;
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float t = acc[i][j][4 * qd + c];
                        if constexpr (ACT == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                        else if constexpr (ACT == MMDM_EPI_BIAS_SILU) t = silu(t);
                        v[c] = t;
                    }
                    if constexpr (SPLIT) {
                        h16x4 oh, ol;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const _Float16 t = mmdm_split_hi(v[c]); oh[c] = t; ol[c] = mmdm_split_lo(v[c], t); }
                        if (rok && col < p.N) {
                            _Float16* cp = static_cast<_Float16*>(p.C) + (size_t)row * p.ldc + col;
                            *reinterpret_cast<h16x4*>(cp) = oh;
                            *reinterpret_cast<h16x4*>(cp + p.pc) = ol;
                        }
                    }
                    if constexpr (SEC) {
                        bf16x4 o1, o2, o3;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            o1[c] = (__bf16)v[c];
                            const float r1 = v[c] - (float)o1[c];
                            o2[c] = (__bf16)r1;
                            o3[c] = (__bf16)(r1 - (float)o2[c]);
                        }
                        if (rok && col < p.N) {
                            {
                                if (col < p.p2_cols) {
                                    __bf16* cp = p.P2 + (size_t)(row + p.row0) * p.ld2 + col;
                                    *reinterpret_cast<bf16x4*>(cp) = o1;
                                    *reinterpret_cast<bf16x4*>(cp + p.p2_plane) = o2;
                                    *reinterpret_cast<bf16x4*>(cp + 2 * p.p2_plane) = o3;
                                }
                            }
                        }
                    }
                    if constexpr (!SPLIT) {
                        if (col < p.N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, voffC + i * 32 * p.ldc * 4 + (j * 32 + 8 * qd) * 4, 0, 0);
                    }
                }
        }
    };
    auto finish_out = [&](auto act_c) {
        if (p.out_split) {
            if (p.P2) finish(act_c, std::true_type{}, std::true_type{});
            else finish(act_c, std::true_type{}, std::false_type{});
        } else {
            if (p.P2) finish(act_c, std::false_type{}, std::true_type{});
            else finish(act_c, std::false_type{}, std::false_type{});
        }
    };
    if (p.epilogue == MMDM_EPI_BIAS_GELU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_GELU>{});
    else if (p.epilogue == MMDM_EPI_BIAS_SILU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_SILU>{});
    else finish_out(std::integral_constant<int, MMDM_EPI_BIAS>{});
}
#endif

#ifndef MMDM_ST_AUX
#define MMDM_ST_AUX 2          // cache policy of the transposed epilogue's stores (buffer_store aux: 0 = write-back, 2 = nt): see gemm_f32 / gemm_bf16
#endif
#ifndef MMDM_SPLIT_TST_DEFAULT
#define MMDM_SPLIT_TST_DEFAULT 1
#endif
int g_split_tst = MMDM_SPLIT_TST_DEFAULT;      // mmdm_diag_set "split_tst": 0 = the direct (row-per-lane) epilogue

#if defined(__HIP_DEVICE_COMPILE__)
// Transposed epilogue of the packed-W kernel (four waves side by side, each TM x 1 MFMA tiles: BN = 128).  In the D^T map a lane owns an
// output ROW: the direct form stores 16 bytes (fp32) or two times 8 bytes (the two fp16 planes of FFN-1's output) of 32 different rows per
// instruction.  Here the workgroup's tile goes through an XOR-swizzled image [rows][128 columns] (per plane) in the idle A stages and leaves
// as whole rows; the residual / PE rows of the fp32 form are read the same way -- whole lines, requested before the image is written -- and
// added last, (b + sum_k a_k w_k) + r as in the direct form.  Same values, same arithmetic: bit-identical results (gemm_bf16.hip has the
// measurement: +50 % on the bf16-output projections).  Returns false (nothing done) for what it does not cover: a second plane output, a
// ragged last column tile.
template <int TM, int BM, int NWAVES>
__device__ __forceinline__ bool split_finish_t(const SArgs& p, f32x16 (&acc)[TM][1], int m0, int n0, int wn, int lane, float* smem, int smem_bytes) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int BN = 32 * NWAVES;
    static_assert(BN == 128 && BM == 32 * TM, "four waves side by side, each TM x 1 tiles of 32 x 32");
    if (p.P2 || n0 + BN > p.N || !p.tst || (p.out_split && (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE))) return false;
    const int l31 = lane & 31, lh = lane >> 5, wave = wn;
    __builtin_amdgcn_s_barrier();                             // every wave has left the K loop: the A stages are free
    const bool ext = p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE;
    const int rows_here = min(p.M - m0, BM);
    char* img = reinterpret_cast<char*>(smem);
    auto finish = [&](auto act_c, auto split_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool SPLIT = decltype(split_c)::value;
        constexpr int EBO = SPLIT ? 2 : 4, NOP = SPLIT ? NPL : 1;               // bytes per element, output planes
        constexpr int RB = BN * EBO, CPR = RB / 16;                            // bytes / 16-byte chunks per image row (16 or 32)
        constexpr int FIT = (3 * NPL * BM * 16 * 4) / (32 * RB * NOP);         // 32-row tiles the three A stages hold
        constexpr int RPP = FIT >= TM ? TM : (FIT >= 2 ? 2 : 1);               // row tiles per phase
        static_assert(FIT >= 1 && TM % RPP == 0, "image does not fit the A stages");
        constexpr int LPR = CPR, RPI = 64 / LPR, ROWS = 32 * RPP, RPW = ROWS / NWAVES;
        static_assert(RPW % RPI == 0, "rows of a phase must divide over the waves' store instructions");
        constexpr int PIMG = ROWS * RB;                                        // bytes of one plane's image
        const int rr = lane / LPR, rc = lane % LPR;
        const size_t cplane = SPLIT ? p.pc * 2 : 0;                            // byte stride between output planes
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(p.C) + (size_t)m0 * p.ldc * EBO, 0,
                                                                             SPLIT ? 0xffffffffu : (unsigned)(rows_here * p.ldc * EBO), 0x00020000);
#pragma unroll
        for (int ph = 0; ph < TM / RPP; ++ph) {
            f32x4 rq[SPLIT ? 1 : RPW / RPI];
            if constexpr (!SPLIT) {
                if (ext) {
#pragma unroll
                    for (int k = 0; k < RPW / RPI; ++k) {
                        const int ir = wave * RPW + k * RPI + rr;
                        const int rowc = min(m0 + ph * ROWS + ir, p.M - 1);
                        const int er = p.epilogue == MMDM_EPI_BIAS_PE ? (rowc + p.row0) % p.period : rowc;
                        rq[k] = *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + n0 + rc * 4);
                    }
                }
            }
#pragma unroll
            for (int i = ph * RPP; i < (ph + 1) * RPP; ++i) {
                const int ir = (i - ph * RPP) * 32 + l31;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float t = acc[i][0][4 * qd + c];
                        if constexpr (ACT == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                        else if constexpr (ACT == MMDM_EPI_BIAS_SILU) t = silu(t);
                        v[c] = t;
                    }
                    const int cb = (wn * 32 + 8 * qd + 4 * lh) * EBO;           // byte column inside the image row
                    char* dst = img + ir * RB + (((cb >> 4) ^ (ir & 15)) << 4) + (cb & 15);
                    if constexpr (SPLIT) {
                        h16x4 oh, ol;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const _Float16 t = mmdm_split_hi(v[c]); oh[c] = t; ol[c] = mmdm_split_lo(v[c], t); }
                        *reinterpret_cast<h16x4*>(dst) = oh;
                        *reinterpret_cast<h16x4*>(dst + PIMG) = ol;
                    } else {
                        *reinterpret_cast<f32x4*>(dst) = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int k = 0; k < RPW / RPI; ++k) {
                const int ir = wave * RPW + k * RPI + rr;
                const int trow = ph * ROWS + ir;
#pragma unroll
                for (int pl = 0; pl < NOP; ++pl) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(img + pl * PIMG + ir * RB + ((rc ^ (ir & 15)) << 4));
                    if constexpr (!SPLIT) { if (ext) v += rq[k]; }
                    if constexpr (SPLIT) {
                        if (m0 + trow < p.M)            // (the plane stride defeats the resource's row clipping: explicit, wave-divergent only in the last row tile)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (unsigned)(trow * p.ldc * EBO + n0 * EBO + rc * 16), (unsigned)(pl * cplane), MMDM_ST_AUX);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, trow * p.ldc * EBO + n0 * EBO + rc * 16, 0, MMDM_ST_AUX);
                    }
                }
            }
            if (ph + 1 < TM / RPP) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    };
    auto finish_out = [&](auto act_c) {
        if (p.out_split) finish(act_c, std::true_type{});
        else finish(act_c, std::false_type{});
    };
    if (p.epilogue == MMDM_EPI_BIAS_GELU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_GELU>{});
    else if (p.epilogue == MMDM_EPI_BIAS_SILU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_SILU>{});
    else finish_out(std::integral_constant<int, MMDM_EPI_BIAS>{});
    (void)smem_bytes;
    return true;
}
#endif

// DIAG: the timing-ablation switches (tools/gemm_split_bench.py ABL=) exist in a second instantiation only -- as runtime tests inside the K loop
// they split its basic block and the sched_group_barrier pipeline with it.
template <int TM_, int TN_, bool DIAG = false>
__global__ __launch_bounds__((SCfg<TM_, TN_>::THREADS)) void gemm_split_kernel(SArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)          // the buffer-resource type of the LDS-DMA / buffer-store builtins exists in the device pass only
    using C_ = SCfg<TM_, TN_>;
    const int ablate = DIAG ? p.ablate : 0;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [2][NPL][BM*16]
    float* Bs = smem + 2 * C_::A_FLOATS;           // [2][NPL][BN*16]

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

    // LDS-DMA pieces, buffer-addressed (see gemm_f32.hip): one resource per operand based at the tile's first row, a per-lane byte
    // offset fixed for the tile (plane, row, swizzled chunk) and a scalar offset that walks K.
    constexpr int UA = C_::NA / C_::NWAVES;            // pieces u < UA are A pieces for every wave
    static_assert(C_::NA % C_::NWAVES == 0, "A pieces must divide evenly over the waves");
    int voff[C_::NI], dst[C_::NI];
#pragma unroll
    for (int u = 0; u < C_::NI; ++u) {
        const int pq = wave + C_::NWAVES * u;
        const int prow = lane >> 2, pc = lane & 3;
        const bool isa = u < UA;
        const int pl = isa ? pq / C_::NAP : (pq - C_::NA) / C_::NBP;          // plane
        const int pp = isa ? pq % C_::NAP : (pq - C_::NA) % C_::NBP;          // piece inside the plane
        const int trow = 16 * pp + prow;
        const int gch = pc ^ ((trow >> 2) & 3);
        if (isa) {
            const int rel = min(trow, p.M - 1 - m0);
            voff[u] = (int)(((size_t)pl * p.pa + (size_t)rel * p.lda) * 2) + 16 * gch;
            dst[u] = pl * C_::A_PLANE + 16 * pp * 16;
        } else {
            const int rel = min(trow, p.N - 1 - n0);
            voff[u] = (int)(((size_t)pl * p.pw + (size_t)rel * p.ldw) * 2) + 16 * gch;
            dst[u] = 2 * C_::A_FLOATS + pl * C_::B_PLANE + 16 * pp * 16;
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A + (size_t)m0 * p.lda), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.W + (size_t)n0 * p.ldw), 0, 0xffffffff, 0x00020000);
    int koff = 0;
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < C_::NI; ++u) {
            const int boff = u < UA ? buf * C_::A_FLOATS : buf * C_::B_FLOATS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(u < UA ? rsA : rsW, (lptr_t)(smem + dst[u] + boff), 16, voff[u], koff, 0, 0);
        }
        koff += 64;
    };

    // The hi accumulators start as the bias, the lo accumulators at zero; after the loop  (b + sum hi) + (sum lo) / 2048,  then the residual /
    // PE tile, (..) + r -- the reference's `x + linear(..)` order and the order of every fp32-split kernel, so a row's bits do not depend on
    // the kernel that produced it.
    f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) accl[i][j][e] = 0.f;
    {
        f32x4 bv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int colc = min(n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh, p.N - 4);
                bv[j][qd] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] = bv[j][qd][c];
    }

    const int nkt = p.K / BK;
    const int sw = (l31 >> 2) & 3;
    const int a_row = (wm * (32 * TM) + l31) * 16;
    const int b_row = (wn * (32 * TN) + l31) * 16;
    // Two LDS stages, software-pipelined inside the step (round 2; the loop used to run barrier -> 9 DMA issues -> fragment reads ->
    // MFMAs in sequence, with both waves of a SIMD at the same point: ~20 % of every step with an idle matrix pipe):
    //   * the fragments of k-block 1 are read behind the first MFMA of k-block 0, those of the NEXT stage's k-block 0 behind the 37th
    //     MFMA of the step (two register sets): no MFMA waits on LDS;
    //   * the step's wait + barrier sits before the last 12 MFMAs: by then every wave has read the whole current stage, so the stage
    //     after next is requested right there, its 9 buffer-addressed DMA pieces placed one by one behind those MFMAs.
    // The order of the MFMAs on an accumulator is k-block by k-block (lo: ah*wl, then al*wh).
    h16x8 f0a[NPL][TM], f0b[NPL][TN], f1a[NPL][TM], f1b[NPL][TN];
    auto rd = [&](int buf, int kb, h16x8 (&af)[NPL][TM], h16x8 (&bf)[NPL][TN]) {
        const float* Ac = As + buf * C_::A_FLOATS + a_row;
        const float* Bc = Bs + buf * C_::B_FLOATS + b_row;
        const int cg = 4 * ((2 * kb + lh) ^ sw);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(Ac + pl * C_::A_PLANE + i * 32 * 16 + cg));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(Bc + pl * C_::B_PLANE + j * 32 * 16 + cg));
        }
    };
    // three product terms: ah*wl and al*wh into the lo accumulators, ah*wh into the hi ones; each term walks all TM x TN accumulators so
    // dependent MFMAs are TM*TN apart
    auto mm = [&](auto t0c, auto t1c, const h16x8 (&af)[NPL][TM], const h16x8 (&bf)[NPL][TN]) {
        constexpr int t0 = decltype(t0c)::value, t1 = decltype(t1c)::value;
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (t < 2) accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[PB[t]][j], af[PA[t]][i], accl[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[PB[t]][j], af[PA[t]][i], acc[i][j], 0, 0, 0);
                }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I3 = std::integral_constant<int, 3>;
    constexpr int NRD = NPL * (TM + TN), NT = TM * TN;        // fragment reads per k-block, MFMAs per term
    constexpr int NPAIR = C_::NI < 2 * NT - 1 ? C_::NI : 2 * NT - 1;      // DMA pieces that get an MFMA of their own to hide behind
    stage(0);
    if (nkt > 1) { stage(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_::NI) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(0, 0, f0a, f0b);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (!(ablate & 2)) rd(cur, 1, f1a, f1b);
        mm(I0{}, I3{}, f0a, f0b);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT - 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(I0{}, I1{}, f1a, f1b);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // stage kt+1 landed (stage kt+2 is requested below)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(ablate & 16)) __builtin_amdgcn_s_barrier();        // every wave has read all of stage kt
            __builtin_amdgcn_sched_barrier(0);
            if (!(ablate & 2)) rd(cur ^ 1, 0, f0a, f0b);
            if (kt + 2 < nkt && !(ablate & 1)) stage(cur);           // (ablate bits: timing experiments, tools/gemm_split_bench.py)
        }
        mm(I1{}, I3{}, f1a, f1b);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 1);
#pragma unroll
        for (int u = 0; u < NPAIR; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); }
        __builtin_amdgcn_sched_group_barrier(0x010, C_::NI - NPAIR, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]), "+v"(accl[i][j]));
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_fmaf(accl[i][j][e], MMDM_SPLIT_INV, acc[i][j][e]);
        }

    if (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rowc = min(m0 + wm * (32 * TM) + i * 32 + l31, p.M - 1);
            const int er = p.epilogue == MMDM_EPI_BIAS_PE ? (rowc + p.row0) % p.period : rowc;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int colc = min(n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh, p.N - 4);
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + colc);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] += rv[c];
                }
        }
    }
    split_finish<TM, TN, BM>(p, acc, m0, n0, wm, wn, l31, lh);
#endif
}

template <int TM_, int TN_>
int launch(SArgs a, hipStream_t st) {
    using C_ = SCfg<TM_, TN_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    mmdm_note_gemm("gemm_split<%d,%d>", TM_, TN_);
    if (a.ablate & 19) hipLaunchKernelGGL((gemm_split_kernel<TM_, TN_, true>), dim3(a.mt * a.nt), dim3(C_::THREADS), C_::SMEM_BYTES, st, a);
    else hipLaunchKernelGGL((gemm_split_kernel<TM_, TN_>), dim3(a.mt * a.nt), dim3(C_::THREADS), C_::SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_split");
}

template <int TM_, int TN_>
int set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<TM_, TN_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SCfg<TM_, TN_>::SMEM_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<TM_, TN_, true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, SCfg<TM_, TN_>::SMEM_BYTES);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_split): %s", hipGetErrorString(e));
    return MMDM_OK;
}


// ---- W straight from global memory (round 2, last experiment of the round) ------------------------------------------------------------
// Timing ablations of the kernel above (ABL bits, tools/gemm_split_bench.py) showed that the W side of the LDS traffic -- a third of the
// LDS-DMA pieces and half of the fragment reads -- costs 35 % of the layer shapes' rate (QKV 177 -> 238 TFLOP/s without it).  Weights
// are static, so mmdm_prepare stores them in FRAGMENT ORDER (mmdm_split_pack_weight): block (plane, 32 rows, 16 k) is the 1 KiB that one
// wave-wide 16-byte load delivers as the MFMA's B operand, lane (l31, lh) <- row 32*nb + l31, k = 16*kb + 8*lh .. +8.  The kernel then
// fetches its B fragments with buffer_load_dwordx4 one whole step ahead (two register sets that swap every step), and LDS carries A only.
// Same six-term order per accumulator: bit-identical to gemm_split_kernel.
template <int TM_, int TN_, bool TL = false, int NV1_ = 0>
__global__ __launch_bounds__((SCfg<TM_, TN_>::THREADS), 2) void gemm_splitw_kernel(SArgs p) {      // two waves per SIMD: 256 registers (two workgroups per CU)
#if defined(__HIP_DEVICE_COMPILE__)
    using C_ = SCfg<TM_, TN_>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN;
    constexpr int NIA = C_::NA / C_::NWAVES;           // LDS-DMA pieces per wave and step (A only)
    constexpr int NLB = 2 * NPL * TN;                  // B-fragment loads per wave and step
    static_assert(C_::NA % C_::NWAVES == 0, "A pieces must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                  // [3 stages][NPL planes][BM*16]

    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    // tiles that run together on an XCD (its chunk of consecutive indices) form blocks of GM m-tiles x 8 n-tiles, so both the A rows and the
    // W blocks they stream are shared in that XCD's L2 (with n fastest, a 24-tile-wide QKV launch had every W block used by < 3 workgroups at a time)
    int mi, ni;
    {
        const int GM = p.ablate & 32 ? 1 : 8;
        const int per_g = GM * p.nt, g = swz / per_g, rem = swz - g * per_g;
        const int gm = min(GM, p.mt - g * GM);               // rows in this (possibly last, shorter) group
        ni = rem / gm; mi = g * GM + rem - ni * gm;
    }
    const int m0 = mi * BM;
    const int n0 = ni * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C_::WGN, wn = wave % C_::WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nkt = p.K / BK;

    int voff[NIA], dst[NIA];
#pragma unroll
    for (int u = 0; u < NIA; ++u) {
        const int pq = wave + C_::NWAVES * u;
        const int prow = lane >> 2, pc = lane & 3;
        const int pl = pq / C_::NAP, pp = pq % C_::NAP;
        const int trow = 16 * pp + prow;
        const int gch = pc ^ ((trow >> 2) & 3);
        const int rel = min(trow, p.M - 1 - m0);
        voff[u] = (int)(((size_t)pl * p.pa + (size_t)rel * p.lda) * 2) + 16 * gch;
        dst[u] = pl * C_::A_PLANE + 16 * pp * 16;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A + (size_t)m0 * p.lda), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.W), 0, 0xffffffff, 0x00020000);
    auto stage_part = [&](int buf, int kt, auto u0c, auto u1c) {      // kt clamped: the loop stays branch-free, a surplus request re-reads the last tile
        const int koff = min(kt, nkt - 1) * 64;
#pragma unroll
        for (int u = decltype(u0c)::value; u < decltype(u1c)::value; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + dst[u] + buf * C_::A_FLOATS), 16, voff[u], koff, 0, 0);
    };
    auto stage = [&](int buf, int kt) { stage_part(buf, kt, std::integral_constant<int, 0>{}, std::integral_constant<int, NIA>{}); };
    // packed W: byte offset of block (plane, nb, kb16) = plane*pw*2 + (nb*(K/16) + kb16)*1024; this wave owns nb = n0/32 + wn*TN + j
    int voffW[NPL][TN];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j) voffW[pl][j] = (int)((size_t)pl * p.pw * 2) + ((n0 >> 5) + wn * TN + j) * (p.K >> 4) * 1024 + lane * 16;
    auto ldb = [&](int kt, h16x8 (&b0)[NPL][TN], h16x8 (&b1)[NPL][TN]) {
        const int so = min(kt, nkt - 1) * 2048;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) b0[pl][j] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW[pl][j], so, 0));
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[pl][j] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW[pl][j] + 1024, so, 0));
    };

    f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) accl[i][j][e] = 0.f;
    {
        f32x4 bv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int colc = min(n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh, p.N - 4);
                bv[j][qd] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] = bv[j][qd][c];
    }

    const int sw = (l31 >> 2) & 3;
    const int a_row = (wm * (32 * TM) + l31) * 16;
    h16x8 f0a[NPL][TM], f1a[NPL][TM];
    h16x8 bx0[NPL][TN], bx1[NPL][TN], by0[NPL][TN], by1[NPL][TN];
    auto rda = [&](int buf, int kb, h16x8 (&af)[NPL][TM]) {
        const float* Ac = As + buf * C_::A_FLOATS + a_row;
        const int cg = 4 * ((2 * kb + lh) ^ sw);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = __builtin_bit_cast(h16x8, *reinterpret_cast<const f32x4*>(Ac + pl * C_::A_PLANE + i * 32 * 16 + cg));
    };
    // three product terms: ah*wl and al*wh into the lo accumulators, ah*wh into the hi ones; each term walks all TM x TN accumulators so
    // dependent MFMAs are TM*TN apart
    auto mm = [&](auto t0c, auto t1c, const h16x8 (&af)[NPL][TM], const h16x8 (&bf)[NPL][TN]) {
        constexpr int t0 = decltype(t0c)::value, t1 = decltype(t1c)::value;
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (t < 2) accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[PB[t]][j], af[PA[t]][i], accl[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[PB[t]][j], af[PA[t]][i], acc[i][j], 0, 0, 0);
                }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I3 = std::integral_constant<int, 3>;
    constexpr int NRDA = NPL * TM, NT = TM * TN;
    constexpr int MB = 4 * NT, MA = 2 * NT;                     // MFMAs of a step before / after its barrier: k-block 0 (three terms) + ah*wl of k-block 1 | the other two
    // one K step: B fragments (b0, b1) of this step are in registers (or on their way: the compiler counts them), (n0_, n1_) receive the next step's
    unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int slot) {              // s_memtime (shader clock); the read drains lgkmcnt, so stamps sit where that is due anyway
        if constexpr (TL) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_readcyclecounter();
            if (slot >= 0) tsum[slot] += t - tprev;
            tprev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Three A stages: the stage after next is requested DURING the step, not after its barrier.  The in-kernel timeline (tools/split_timeline.py)
    // of the two-stage form showed why: all eight waves leave the barrier together and push their LDS-DMA requests into the CU's one address
    // unit at once (16 cycles per 1-KiB instruction); a wave whose request cannot issue cannot issue the MFMAs behind it either -- the 12 MFMAs
    // after the barrier took 125 cycles each instead of 32, and the waves reached the next barrier 900 cycles apart.  With a third stage
    // buffer (kt+2) % 3 is free from the previous step's barrier on, so the step's VMEM instructions -- next step's B fragments and the A pieces of the
    // stage after next -- are spread one by one over the 4*NT MFMAs before the barrier, NV1 of the A pieces over the 2*NT after it.
    constexpr int NV1 = NV1_ < NIA ? NV1_ : NIA;                 // A pieces requested after the barrier
    constexpr int NV0 = NLB + NIA - NV1;                            // VMEM instructions before it
    constexpr int SP0 = (MB - 1) / NV0 > 0 ? (MB - 1) / NV0 : 1, NP0 = NV0 * SP0 <= MB - 1 ? NV0 : (MB - 1) / SP0;
    constexpr int NP1 = NV1 < MA - 1 ? NV1 : MA - 1;
    auto step = [&](int kt, int cur, int nxt, int nn, h16x8 (&b0)[NPL][TN], h16x8 (&b1)[NPL][TN], h16x8 (&n0_)[NPL][TN], h16x8 (&n1_)[NPL][TN]) {
        stamp(kt == 0 ? -1 : 4);
        rda(cur, 1, f1a);
        ldb(kt + 1, n0_, n1_);
        stage_part(nn, kt + 2, std::integral_constant<int, 0>{}, std::integral_constant<int, NIA - NV1>{});
        mm(I0{}, I3{}, f0a, b0);
        mm(I0{}, I1{}, f1a, b1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRDA, 0);
#pragma unroll
        for (int u = 0; u < NP0; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, SP0, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x010, NV0 - NP0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MB - 1 - SP0 * NP0, 0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV0) : "memory");     // A stage kt+1 landed (requested during step kt-1); this step's requests may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(2);
        __builtin_amdgcn_s_barrier();                                   // every wave has read all of stage kt and sees stage kt+1
        __builtin_amdgcn_sched_barrier(0);
        stamp(3);
        rda(nxt, 0, f0a);
        stage_part(nn, kt + 2, std::integral_constant<int, NIA - NV1>{}, std::integral_constant<int, NIA>{});
        mm(I1{}, I3{}, f1a, b1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NRDA, 1);
#pragma unroll
        for (int u = 0; u < NP1; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); }
        __builtin_amdgcn_sched_group_barrier(0x010, NV1 - NP1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, MA, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    unsigned long long rt0 = 0, ct0 = 0;
    if constexpr (TL) { rt0 = __builtin_amdgcn_s_memrealtime(); ct0 = __builtin_readcyclecounter(); }
    stage(0, 0);
    ldb(0, bx0, bx1);
    stage(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIA) : "memory");
    __builtin_amdgcn_s_barrier();
    rda(0, 0, f0a);
    int cur = 0;
    for (int kt = 0; kt < nkt; kt += 2) {                 // nkt is even (host check): the two B register sets swap roles every step
        const int c1 = cur == 2 ? 0 : cur + 1, c2 = c1 == 2 ? 0 : c1 + 1;
        step(kt, cur, c1, c2, bx0, bx1, by0, by1);
        step(kt + 1, c1, c2, cur, by0, by1, bx0, bx1);
        cur = c2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus requests of the last steps
    if constexpr (TL) {
        stamp(4);
        if (p.tl && lane == 0) {
            unsigned long long* o = p.tl + ((size_t)blockIdx.x * C_::NWAVES + wave) * 8;
#pragma unroll
            for (int i = 0; i < 5; ++i) o[i] = tsum[i];
            o[5] = (unsigned long long)nkt;
            o[6] = __builtin_amdgcn_s_memrealtime() - rt0;       // 100 MHz reference ticks over the loop
            o[7] = __builtin_readcyclecounter() - ct0;           // s_memtime ticks over the same stretch: their ratio is the shader clock
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]), "+v"(accl[i][j]));
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = __builtin_fmaf(accl[i][j][e], MMDM_SPLIT_INV, acc[i][j][e]);
        }

    if constexpr (TN == 1 && C_::WGM == 1 && C_::WGN == 4) {
        if (split_finish_t<TM, BM, C_::NWAVES>(p, acc, m0, n0, wn, lane, smem, 3 * C_::A_FLOATS * 4)) return;
    }
    if (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rowc = min(m0 + wm * (32 * TM) + i * 32 + l31, p.M - 1);
            const int er = p.epilogue == MMDM_EPI_BIAS_PE ? (rowc + p.row0) % p.period : rowc;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int colc = min(n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh, p.N - 4);
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + colc);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] += rv[c];
                }
        }
    }
    split_finish<TM, TN, BM>(p, acc, m0, n0, wm, wn, l31, lh);
#endif
}

template <int TM_, int TN_, int NV1_ = 0>
int launch_w(SArgs a, hipStream_t st) {
    using C_ = SCfg<TM_, TN_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    mmdm_note_gemm("gemm_splitw<%d,%d>", TM_, TN_);
    if (a.tl) hipLaunchKernelGGL((gemm_splitw_kernel<TM_, TN_, true, NV1_>), dim3(a.mt * a.nt), dim3(C_::THREADS), 3 * C_::A_FLOATS * 4, st, a);
    else hipLaunchKernelGGL((gemm_splitw_kernel<TM_, TN_, false, NV1_>), dim3(a.mt * a.nt), dim3(C_::THREADS), 3 * C_::A_FLOATS * 4, st, a);
    return mmdm_check_launch("gemm_splitw");
}

template <int TM_, int TN_, int NV1_ = 0>
int set_attr_w() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_splitw_kernel<TM_, TN_, false, NV1_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SCfg<TM_, TN_>::A_FLOATS * 4);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_splitw_kernel<TM_, TN_, true, NV1_>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SCfg<TM_, TN_>::A_FLOATS * 4);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_splitw): %s", hipGetErrorString(e));
    return MMDM_OK;
}

// [NPL][N][K] planes -> fragment order (see gemm_splitw_kernel); one thread per 16-byte chunk
__global__ void pack_split_w_kernel(const _Float16* __restrict__ in, int ldw, size_t in_plane, _Float16* __restrict__ out, size_t out_plane, int N, int K) {
    const size_t per_plane = (size_t)N * K / 8;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < NPL * per_plane; c += (size_t)gridDim.x * blockDim.x) {
        const int pl = (int)(c / per_plane);
        const size_t o = c % per_plane;                       // chunk index inside the plane: ((nb*KB16 + kb16)*64 + lh*32 + l31)
        const int ln = (int)(o & 63), l31 = ln & 31, lh = ln >> 5;
        const size_t blk = o >> 6;
        const int kb16 = (int)(blk % (size_t)(K >> 4)), nb = (int)(blk / (size_t)(K >> 4));
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + pl * in_plane + (size_t)(nb * 32 + l31) * ldw + kb16 * 16 + lh * 8);
        *reinterpret_cast<f32x4*>(out + pl * out_plane + o * 8) = v;
    }
}

// x -> two fp16 planes out[0], out[plane]: x ~= out0 + out1 / 2048 (mmdm_split2, kernels.h)
__global__ void split_kernel(const float* __restrict__ in, _Float16* __restrict__ out, size_t n, size_t plane) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        _Float16 h, l;
        mmdm_split2(in[i], h, l);
        out[i] = h;
        out[plane + i] = l;
    }
}

int g_split_cfg = -1;
int g_split_ablate = 0;
unsigned long long* g_split_tl = nullptr;

}  // namespace

int mmdm_gemm_split_init(void) {
    int rc;
    if ((rc = set_attr<22, 22>())) return rc;
    if ((rc = set_attr<42, 22>())) return rc;
    if ((rc = set_attr<22, 21>())) return rc;
    if ((rc = set_attr<24, 22>())) return rc;
    if ((rc = set_attr_w<22, 21>())) return rc;
    if ((rc = set_attr_w<12, 41>())) return rc;
    if ((rc = set_attr_w<14, 41>())) return rc;
    return MMDM_OK;
}

// diagnostics of this translation unit (mmdm_diag_set): tile override, ablation bits, timeline buffer (8 u64 per wave of the next packed launches)
bool mmdm_diag_gemm_split(const char* key, long long v) {
    if (!strcmp(key, "split_cfg")) g_split_cfg = (int)v;
    else if (!strcmp(key, "split_tst")) g_split_tst = (int)v;
    else if (!strcmp(key, "split_ablate")) g_split_ablate = (int)v;
    else if (!strcmp(key, "split_timeline")) g_split_tl = reinterpret_cast<unsigned long long*>((uintptr_t)v);
    else return false;
    return true;
}

extern "C" int mmdm_f32_split(const float* in, void* out, int64_t n, int64_t plane_stride, void* stream) {
    if (n <= 0) return MMDM_OK;
    if (!in || !out || plane_stride < n) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_f32_split: bad arguments");
    hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), in, static_cast<_Float16*>(out), (size_t)n, (size_t)plane_stride);
    return mmdm_check_launch("f32_split");
}

extern "C" int mmdm_linear_split(const void* A, int lda, int64_t a_plane, const void* W, int ldw, int64_t w_plane, const float* bias, void* C, int ldc,
                                 int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_split_ex(A, lda, a_plane, W, ldw, w_plane, bias, C, ldc, c_plane, out_split, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, 0, stream);
}

extern "C" int mmdm_linear_split_packed(const void* A, int lda, int64_t a_plane, const void* Wp, int64_t w_plane, const float* bias, void* C, int ldc,
                                        int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_split_ex(A, lda, a_plane, Wp, 0, w_plane, bias, C, ldc, c_plane, out_split, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, 0, stream);
}

extern "C" int mmdm_split_pack_weight(const void* W, int ldw, int64_t w_plane, void* out, int64_t out_plane, int N, int K, void* stream) {
    if (N == 0) return MMDM_OK;
    if (!W || !out || N < 0 || K <= 0 || (N & 31) || (K & 15) || ldw < K || (ldw & 7) || (w_plane & 7) || out_plane < (int64_t)N * K || (out_plane & 7))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_split_pack_weight: needs N %% 32 == 0, K %% 16 == 0, 16-byte aligned rows / planes (N=%d K=%d ldw=%d)", N, K, ldw);
    if ((reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_split_pack_weight: unaligned pointer");
    hipLaunchKernelGGL(pack_split_w_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const _Float16*>(W), ldw, (size_t)w_plane,
                       static_cast<_Float16*>(out), (size_t)out_plane, N, K);
    return mmdm_check_launch("split_pack_weight");
}

// ldw == 0: W is in fragment order (mmdm_split_pack_weight); served by gemm_splitw_kernel
int mmdm_linear_split_ex(const void* A, int lda, int64_t a_plane, const void* W, int ldw, int64_t w_plane, const float* bias, void* C, int ldc,
                         int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period,
                         void* planes2, int ld2, int64_t plane2_stride, int planes2_cols, void* stream) {
    const bool packed = ldw == 0;
    if (packed) {
        if ((N & 63) || (K & 63)) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split_packed: needs N %% 64 == 0 and K %% 64 == 0 (N=%d K=%d)", N, K);
        ldw = K;
    }
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
    if ((uint64_t)a_plane * 4 + 512ull * (uint64_t)lda >= (1ull << 32) || (uint64_t)w_plane * 4 + 512ull * (uint64_t)ldw >= (1ull << 32))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split: operand planes beyond the 4 GB reach of a tile's buffer offsets");
    // plane output: the lo plane is addressed as a 32-bit scalar offset (c_plane * 2 bytes) beside a 32-bit per-lane offset inside the tile's rows
    if (out_split && (uint64_t)c_plane * 2 + 512ull * (uint64_t)ldc >= (1ull << 32))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split: output planes beyond the 4 GB reach of a tile's buffer offsets (c_plane=%lld)", (long long)c_plane);
    SArgs a;
    a.A = static_cast<const _Float16*>(A); a.W = static_cast<const _Float16*>(W); a.bias = bias; a.C = C; a.extra = extra;
    a.pa = (size_t)a_plane; a.pw = (size_t)w_plane; a.pc = (size_t)c_plane;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ld_extra = ld_extra;
    a.M = M; a.N = N; a.K = K; a.epilogue = epilogue; a.period = period > 0 ? period : 1; a.out_split = out_split;
    a.mt = a.nt = 0; a.ablate = g_split_ablate; a.row0 = 0; a.tl = packed ? g_split_tl : nullptr;
    a.P2 = static_cast<__bf16*>(planes2); a.p2_plane = (size_t)plane2_stride; a.p2_cols = planes2_cols; a.ld2 = ld2;
    a.tst = g_split_tst;
    if (planes2 && ((ld2 & 3) || (plane2_stride & 3) || (planes2_cols & 3) || !al16(planes2)))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_split: second output needs 8-byte aligned bf16 rows / planes");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (packed) {
        // measured at M = 19 200 (tools/gemm_split_bench.py, PCFGS): 128x128 tiles of four 128x32 waves, two workgroups per CU (215-221 TFLOP/s on
        // the layer shapes against 182-192 of the plane kernels); 64x128 tiles for the N <= 512 mixer GEMMs; 128x64 when N is not a multiple of 128
        if (N & 127) return launch_w<22, 21>(a, st);
        // small M (the reference's B = 1 call: M = 1196): when the 128 x 128 grid gives fewer tiles than there are CUs, halve the tile (bit-identical results;
        // fp32_split at B = 1, T = 299: 3.70 -> 3.49 ms/step)
        const bool few = (long)((M + 127) / 128) * (N / 128) < 256;
        // (M <= 64: the conditioning projections of the low-precision handles, one 64-row tile deep -- a weight-streaming launch)
        if (g_split_cfg == 5 || (g_split_cfg != 6 && (N <= 512 || few || M <= 64))) return launch_w<12, 41>(a, st);
        // (a main launch of whole rounds plus a 64x128 remainder, as the plane kernels do for N <= 1024, was measured: no gain or slower -- the
        // kernel runs against the power-managed clock, not against the round count: tools/split_timeline.py reads 1.5 GHz inside the loop)
        return launch_w<14, 41>(a, st);
    }
    switch (g_split_cfg) {
        case 0: return launch<22, 22>(a, st);
        case 1: return launch<42, 22>(a, st);
        case 2: return launch<22, 21>(a, st);
        case 4: return launch<24, 22>(a, st);
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
    b.C = out_split ? static_cast<void*>(static_cast<_Float16*>(C) + (size_t)M1 * ldc) : static_cast<void*>(static_cast<float*>(C) + (size_t)M1 * ldc);
    if (ext && epilogue == MMDM_EPI_BIAS_RESID) b.extra = extra + (size_t)M1 * ld_extra;
    return launch<22, 21>(b, st);
}
