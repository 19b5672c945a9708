// bf16-operand linear layer for gfx950 (BASELINE configs[4] "bf16 path"): C = A W^T + b, A and W bf16, fp32 accumulate on
// v_mfma_f32_32x32x16_bf16, fused epilogues, fp32 or bf16 output.
//
// Same structure as gemm_glds_kernel (gemm_f32.hip): a 64-byte LDS row holds 32 bf16 instead of 16 floats, so the LDS image,
// the 1-KiB LDS-DMA pieces and the XOR swizzle (16-byte chunk c of row r at c ^ ((r >> 2) & 3)) are byte-for-byte the same;
// one ds_read_b128 per operand now feeds ONE MFMA of K = 16 (lane (r, h) holds k = 8h .. 8h+7 of MFMA k-block kb: chunk 2kb + h).
// Operands are swapped (D^T = W A^T) so that the output row sits on the lane and four consecutive columns in consecutive
// accumulator registers: bias / residual loads and result stores are 16-byte (fp32) or 8-byte (bf16) accesses.
//
// ET = 1: fp8 operands (OCP e4m3, BASELINE configs[4] "fp8 MFMA QKV/FFN GEMMs") through the SAME loop: a 64-byte LDS row then holds 64
// fp8 values, every 16-byte fragment read feeds two v_mfma_f32_32x32x16_fp8_fp8 (its two 8-byte halves; the k order inside a row is
// permuted identically for both operands), and the result is de-quantised in the epilogue,
//   C[m][n] = acc[m][n] * a_scale[m] * w_scale[n] + bias[n] (+ residual),
// with a per-row activation scale (written by the quantising producer: AdaLN sees the whole row) and a per-output-channel weight
// scale (mmdm_prepare).  Half the operand bytes of the bf16 form per multiply-add; the non-scaled fp8 MFMA issues at the bf16 rate.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"
#include "gemm_fp8p.h"

bool mmdm_diag_gemm_fp8p(const char* key, long long v);      // gemm_fp8p.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct BArgs {
    const __bf16* A; const __bf16* W; const float* bias; void* C; const float* extra;
    int lda, ldw, ldc, ld_extra;          // element strides
    int M, N, K, epilogue, period, out_bf16;   // out_bf16: 0 fp32, 1 bf16, 2 fp8 e4m3 at unit scale (GELU output feeding the fp8 FFN down-projection)
    int mt, nt;
    const float* a_scale; const float* w_scale;   // fp8 operands: per-row / per-output-channel de-quantisation scales (nullptr = 1)
    float a_const, out_scale;                     // fp8: de-quantisation factor of A when a_scale is null; fp8 output is e4m3(value * out_scale)
    unsigned long long* tl;               // diagnostic (tools/bf16w_timeline.py): per-workgroup {loop shader clocks, loop 100 MHz ticks, whole-kernel ticks, K steps}; null in production
    __bf16* P2; int p2_cols, ld2;         // optional second output: columns [0, p2_cols) also as bf16 (attention Q/K operands)
    int tst;                              // results leave through the workgroup's LDS transposition (bf16_finish_t)
};


template <int TM_, int TN_>
struct BCfg {
    static constexpr int WGM = TM_ / 10, WGN = TN_ / 10, TM = TM_ % 10, TN = TN_ % 10;
    static constexpr int NWAVES = WGM * WGN, THREADS = 64 * NWAVES;
    static constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = 32;          // K step in bf16 elements (64 bytes)
    static constexpr int A_FLOATS = BM * 16, B_FLOATS = BN * 16;                  // LDS sizes in 4-byte units
    static constexpr int SMEM_BYTES = 2 * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int NA = BM / 16, NB = BN / 16;                              // 1-KiB pieces (16 rows x 64 B)
    static constexpr int NI = (NA + NB) / NWAVES;
    static_assert((NA + NB) % NWAVES == 0, "pieces must divide evenly over the waves");
};

__device__ __forceinline__ unsigned pack_fp8x4(f32x4 v) {
    // OCP e4m3, round to nearest even, saturating at +-448 (the conversion instruction does not clamp by itself in the default mode)
    float c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = fminf(fmaxf(v[i], -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
    return (unsigned)w;
}

// Cache policy of the transposed epilogue's stores (aux immediate of buffer_store: 0 = write-back through L2, 2 = nt): whole lines that the
// next kernel streams once.
#ifndef MMDM_BT_AUX
#define MMDM_BT_AUX 2
#endif
#ifndef MMDM_BF16_TST_DEFAULT
#define MMDM_BF16_TST_DEFAULT 1
#endif
int g_bf16_tst = MMDM_BF16_TST_DEFAULT;        // mmdm_diag_set "bf16_tst": 0 = the direct (row-per-lane) epilogue
#ifndef MMDM_FP8P_DEFAULT
#define MMDM_FP8P_DEFAULT 0
#endif
// mmdm_diag_set "fp8p": which packed fp8 launches take the persistent kernel (gemm_fp8p.hip: a tile's epilogue under the next tile's K loop; bit-identical).
// 0 (default) = none: stand-alone the kernel wins 3-5 % on the cross-attention projections (27.5 vs 29.4 us, 48.9 vs 50.4 us at M = 19 200), is even on QKV
// and loses on the GELU epilogue -- and inside the two-stream step it is neutral at B = 16 (9.07 vs 9.06-9.10 ms/step) and costs 0.2 ms at B = 64
// (tools/fp8_step_ab.py; LAB_NOTES.md round 6).  1 = the cross-attention projections (bf16 output, N <= 2048); 2 = every shape it covers (tests, tools/gemm_fp8_bench.py)
int g_fp8p = MMDM_FP8P_DEFAULT;
int g_bf16_lds_pad = 0;                        // mmdm_diag_set "bf16_lds_pad": extra dynamic LDS bytes per packed-W workgroup (occupancy experiments of tools/)

#if defined(__HIP_DEVICE_COMPILE__)
// Transposed epilogue.  In the D^T map a lane owns an output ROW: the direct form below writes bf16 results as 8-byte pieces (fp8: 4-byte) of
// 32 different rows per instruction -- every 128-byte line of C is assembled from 8-16 partial writes of two waves -- and at K = 1024 a tile's
// K loop (8-16 steps) is shorter than that epilogue.  Here the workgroup's tile goes, a few 32-row tiles at a time, through an XOR-swizzled
// image [rows][BN columns] in the (now idle) operand stages, and leaves as whole rows: one instruction stores 64 x 16 contiguous bytes per
// row group.  Values and arithmetic are those of bf16_finish: bit-identical results.  Returns false (nothing done) for the cases it does not
// cover: a second plane output, a ragged last column tile.
// The residual / PE rows of an fp32-output tile that takes the transposed epilogue are added THERE -- (b + sum_k a_k w_k) + r, the order of the
// reference's `x + linear(...)`, read as whole lines -- instead of starting the accumulators: a tile's first MFMA no longer waits for 64 KB
// of residual (plus a vmcnt(0) in front of the operand pipeline).  Same predicate at kernel start and in the epilogue.
__device__ __forceinline__ bool bf16_late_ext(const BArgs& p, int n0, int BN) {
    return p.tst && !p.P2 && n0 + BN <= p.N && p.out_bf16 == 0 && (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE);
}

// SMEM = bytes of dynamic LDS the launch really has (the operand stages): the image of a phase is sized against THAT, not against a constant.
template <int TM, int TN, int ET, int BM, int BN, int NWAVES, int SMEM>
__device__ __forceinline__ bool bf16_finish_t(const BArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, float* smem, const float* sc = nullptr) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if (p.P2 || n0 + BN > p.N) return false;
    const int l31 = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    __builtin_amdgcn_s_barrier();                             // every wave has left the K loop: the operand stages are free
    auto finish = [&](auto act_c, auto out_c) {
        constexpr int ACT = decltype(act_c)::value, OUT = decltype(out_c)::value;
        constexpr int EBO = OUT == 0 ? 4 : (OUT == 1 ? 2 : 1);                 // bytes per output element
        constexpr int RB = BN * EBO, CPR = RB / 16;                            // bytes / 16-byte chunks per image row
        constexpr int NRT = BM / 32;                                           // 32-row tiles of the workgroup tile
        constexpr int FIT = SMEM / (32 * RB);                                  // row tiles the launch's operand stages hold
        constexpr int RPP = FIT >= NRT ? NRT : (FIT >= 4 ? 4 : (FIT >= 2 ? 2 : 1));      // row tiles per phase (a power of two that divides NRT)
        static_assert(FIT >= 1 && NRT % RPP == 0, "image does not fit the operand stages");
        static_assert(32 * RPP * RB <= SMEM, "a phase's image must lie inside the launch's dynamic LDS");
        static_assert(CPR <= 64, "a row of the image is at most one store instruction");
        constexpr int LPR = CPR, RPI = 64 / LPR;                               // lanes per row, rows per store instruction
        constexpr int ROWS = 32 * RPP, RPW = ROWS / NWAVES;                    // rows per phase, rows each wave stores per phase
        static_assert(ROWS % NWAVES == 0 && RPW % RPI == 0, "rows of a phase must divide over the waves' store instructions");
        constexpr int SWM = CPR - 1 < 15 ? CPR - 1 : 15;                       // chunk swizzle mask
        const bool late = OUT == 0 && bf16_late_ext(p, n0, BN);                // residual / PE rows added in phase (2), coalesced
        const bool ext = (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) && !late;
        const int rows_here = min(p.M - m0, BM);
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(p.C) + (size_t)m0 * p.ldc * EBO, 0, rows_here * p.ldc * EBO, 0x00020000);
        char* img = reinterpret_cast<char*>(smem);
        const int rr = lane / LPR, rc = lane % LPR;                            // phase (2): this lane's row inside a store instruction's row group, its 16-byte column
        // fp8: the de-quantisation operands of the workgroup's tile -- a_scale of its BM rows, w_scale and bias of its BN columns -- may sit in LDS
        // (`sc`: [BM | BN | BN] floats behind the operand stages, filled by the kernel's prologue: gemm_bf16w_kernel).  Loaded from global memory
        // where they are used, behind run-time null tests, every (row tile, column quad) group is followed by its own s_waitcnt vmcnt(0): 16 memory
        // latencies in sequence per wave after the K loop (tools/bf16w_timeline.py: 7.0 us of a 15.2 us workgroup on the QKV shape; 4.8 us with
        // the operands in LDS).  (Reading them from LDS in ONE batch ahead of the phases was measured too: 181 registers instead of 168 -- two
        // workgroups per CU instead of three -- not kept; neither was s_setprio 3 for the epilogue: no change.)
#pragma unroll
        for (int ph = 0; ph < NRT / RPP; ++ph) {
            // (0) the residual / PE quads this lane adds in phase (2): requested first, they land behind the image write and the barrier
            f32x4 rq[OUT == 0 ? RPW / RPI : 1];
            if constexpr (OUT == 0) {
                if (late) {
#pragma unroll
                    for (int k = 0; k < RPW / RPI; ++k) {
                        const int ir = wave * RPW + k * RPI + rr;
                        const int rowc = min(m0 + (ph * RPP + ir / 32) * 32 + (ir & 31), p.M - 1);
                        const int er = p.epilogue == MMDM_EPI_BIAS_PE ? rowc % p.period : rowc;
                        rq[k] = *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + n0 + rc * 4);
                    }
                }
            }
            // (1) this wave's row tiles of the phase -> image (values exactly as the direct epilogue forms them).  Column quads outermost: the
            // per-column operands of a quad (w_scale, bias: LDS or global) are read once for the wave's row tiles of the phase, the per-row ones
            // (a_scale) once per row tile up front -- 8 + 4 short waits per wave instead of 48 (rows outermost)
            float sa_i[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rt = wm * TM + i;
                sa_i[i] = 1.f;
                if (rt / RPP != ph) continue;
                const int rowc = min(m0 + rt * 32 + l31, p.M - 1);
                if constexpr (ET == 1) sa_i[i] = sc ? sc[rt * 32 + l31] : (p.a_scale ? p.a_scale[rowc] : p.a_const);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int lc = wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh;               // column inside the tile
                    const int col = n0 + lc;
                    f32x4 add0 = {0.f, 0.f, 0.f, 0.f}, sw4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (ET == 1) {
                        if (sc) {
                            sw4 = *reinterpret_cast<const f32x4*>(sc + BM + lc);
                            add0 = *reinterpret_cast<const f32x4*>(sc + BM + BN + lc);
                        } else {
                            if (p.w_scale) sw4 = *reinterpret_cast<const f32x4*>(p.w_scale + col);
                            if (p.bias) add0 = *reinterpret_cast<const f32x4*>(p.bias + col);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int rt = wm * TM + i;                                // row tile of the workgroup tile (wave-uniform)
                        if (rt / RPP != ph) continue;
                        const int ir = (rt % RPP) * 32 + l31;                      // image row
                        const int row = m0 + rt * 32 + l31, rowc = min(row, p.M - 1);
                        const float sa = sa_i[i];
                        const int er = p.epilogue == MMDM_EPI_BIAS_PE ? rowc % p.period : rowc;
                        f32x4 v, add = add0;
                        if constexpr (ET == 1) {
                            if (ext) add += *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + col);
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float t = acc[i][j][4 * qd + c];
                            if constexpr (ET == 1) t = t * (sa * sw4[c]) + add[c];
                            if constexpr (ACT == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                            else if constexpr (ACT == MMDM_EPI_BIAS_SILU) t = silu(t);
                            v[c] = t;
                        }
                        const int cb = lc * EBO;                                        // byte column inside the image row
                        char* dst = img + ir * RB + (((cb >> 4) ^ (ir & SWM)) << 4) + (cb & 15);
                        if constexpr (OUT == 1) { const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; *reinterpret_cast<bf16x4*>(dst) = o; }
                        else if constexpr (OUT == 2) *reinterpret_cast<unsigned*>(dst) = pack_fp8x4(v * p.out_scale);
                        else *reinterpret_cast<f32x4*>(dst) = v;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // (2) image -> C, whole rows
#pragma unroll
            for (int k = 0; k < RPW / RPI; ++k) {
                const int ir = wave * RPW + k * RPI + rr;
                f32x4 v = *reinterpret_cast<const f32x4*>(img + ir * RB + ((rc ^ (ir & SWM)) << 4));
                if constexpr (OUT == 0) { if (late) v += rq[k]; }
                const int trow = (ph * RPP + ir / 32) * 32 + (ir & 31);                     // row inside the workgroup tile
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, trow * p.ldc * EBO + n0 * EBO + rc * 16, 0, MMDM_BT_AUX);
            }
            if (ph + 1 < NRT / RPP) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                                                // the image is rewritten by the next phase
            }
        }
    };
    auto finish_out = [&](auto act_c) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        if (p.out_bf16 == 1) finish(act_c, I1{});
        else if (p.out_bf16 == 2) finish(act_c, I2{});
        else finish(act_c, I0{});
    };
    if (p.epilogue == MMDM_EPI_BIAS_GELU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_GELU>{});
    else if (p.epilogue == MMDM_EPI_BIAS_SILU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_SILU>{});
    else finish_out(std::integral_constant<int, MMDM_EPI_BIAS>{});
    return true;
}

template <int TM, int TN, int ET, int BM>
__device__ __forceinline__ void bf16_finish(const BArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int l31, int lh) {
    // Activation, output form (fp32 / bf16 / fp8) and the optional bf16 copy are chosen ONCE per tile, outside the element loops (see
    // gemm_f32.hip: with the runtime tests inside them the epilogue was tens of KB of branchy code).  Results leave through buffer stores on
    // a resource that covers the tile's rows of C: rows past M are dropped by the hardware's range check, so the row test -- a divergent
    // branch around every store, crawled through beside co-resident workgroups' MFMA streams (tools/bf16w_timeline.py: half of a
    // residual-shape workgroup's life was outside its K loop) -- is gone; loads of per-row / per-column operands use clamped indices.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto finish = [&](auto act_c, auto out_c, auto sec_c) {
        constexpr int ACT = decltype(act_c)::value, OUT = decltype(out_c)::value;
        constexpr bool SEC = decltype(sec_c)::value;
        constexpr int EBO = OUT == 0 ? 4 : (OUT == 1 ? 2 : 1);                 // bytes per output element
        const bool ext = p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE;
        const int rows_here = min(p.M - m0, BM);
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(p.C) + (size_t)m0 * p.ldc * EBO, 0, rows_here * p.ldc * EBO, 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rrow = wm * (32 * TM) + i * 32 + l31;                    // row inside the tile
            const int row = m0 + rrow;
            const bool rok = row < p.M;
            const int rowc = min(row, p.M - 1);
            float sa = 1.f;
            if constexpr (ET == 1) sa = p.a_scale ? p.a_scale[rowc] : p.a_const;
            const int er = p.epilogue == MMDM_EPI_BIAS_PE ? rowc % p.period : rowc;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int col = n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh;
                    const int colc = min(col, p.N - 4);
                    f32x4 v, add = {0.f, 0.f, 0.f, 0.f}, sw4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (ET == 1) {            // de-quantise, then bias (+ residual / PE row) as the bf16 form's accumulator start does
                        if (p.w_scale) sw4 = *reinterpret_cast<const f32x4*>(p.w_scale + colc);
                        if (p.bias) add = *reinterpret_cast<const f32x4*>(p.bias + colc);
                        if (ext) add += *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + colc);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float t = acc[i][j][4 * qd + c];
                        if constexpr (ET == 1) t = t * (sa * sw4[c]) + add[c];
                        if constexpr (ACT == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                        else if constexpr (ACT == MMDM_EPI_BIAS_SILU) t = silu(t);
                        v[c] = t;
                    }
                    const int voff = (rrow * p.ldc + col) * EBO;
                    if constexpr (OUT == 1 || SEC) {
                        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        if constexpr (OUT == 1) { if (col < p.N) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsC, voff, 0, 0); }
                        if constexpr (SEC) {
                            if (rok && col < p.N && col < p.p2_cols) *reinterpret_cast<bf16x4*>(p.P2 + (size_t)row * p.ld2 + col) = o;
                        }
                    }
                    if constexpr (OUT == 2) { if (col < p.N) __builtin_amdgcn_raw_buffer_store_b32(pack_fp8x4(v * p.out_scale), rsC, voff, 0, 0); }
                    if constexpr (OUT == 0) { if (col < p.N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, voff, 0, 0); }
                }
        }
    };
    auto finish_out = [&](auto act_c) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        if (p.P2) {
            if (p.out_bf16 == 1) finish(act_c, I1{}, std::true_type{});
            else if (p.out_bf16 == 2) finish(act_c, I2{}, std::true_type{});
            else finish(act_c, I0{}, std::true_type{});
        } else {
            if (p.out_bf16 == 1) finish(act_c, I1{}, std::false_type{});
            else if (p.out_bf16 == 2) finish(act_c, I2{}, std::false_type{});
            else finish(act_c, I0{}, std::false_type{});
        }
    };
    if (p.epilogue == MMDM_EPI_BIAS_GELU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_GELU>{});
    else if (p.epilogue == MMDM_EPI_BIAS_SILU) finish_out(std::integral_constant<int, MMDM_EPI_BIAS_SILU>{});
    else finish_out(std::integral_constant<int, MMDM_EPI_BIAS>{});
}
#endif

template <int TM_, int TN_, int ET = 0>
// two 8-wave workgroups per CU = four waves per SIMD (<= 128 registers per wave; the second __launch_bounds__ argument is waves per SIMD): the 256 x 128 tile's 48 KB of LDS allow three, its 64 accumulators two
__global__ __launch_bounds__((BCfg<TM_, TN_>::THREADS), (BCfg<TM_, TN_>::THREADS == 512 ? 4 : 1)) void gemm_bf16_kernel(BArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)          // the buffer-resource type of the LDS-DMA builtin exists in the device pass only
    using C_ = BCfg<TM_, TN_>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [2][BM*16] (4-byte units)
    float* Bs = smem + 2 * C_::A_FLOATS;

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

    // LDS-DMA pieces, buffer-addressed (gemm_f32.hip has the measurement): a resource per operand based at the tile's first row, a
    // per-lane byte offset fixed for the tile, one scalar offset that walks K.
    constexpr int UA = C_::NA / C_::NWAVES;            // pieces u < UA are A pieces for every wave
    static_assert(C_::NA % C_::NWAVES == 0, "A pieces must divide evenly over the waves");
    constexpr int EB = ET == 1 ? 1 : 2;                // bytes per operand element
    int voff[C_::NI], dst[C_::NI];
#pragma unroll
    for (int u = 0; u < C_::NI; ++u) {
        const int pq = wave + C_::NWAVES * u;
        const int prow = lane >> 2, pc = lane & 3;
        const bool isa = u < UA;
        const int trow = 16 * (isa ? pq : pq - C_::NA) + prow;
        const int gch = pc ^ ((trow >> 2) & 3);
        if (isa) {
            voff[u] = min(trow, p.M - 1 - m0) * p.lda * EB + 16 * gch;
            dst[u] = 16 * pq * 16;
        } else {
            voff[u] = min(trow, p.N - 1 - n0) * p.ldw * EB + 16 * gch;
            dst[u] = 2 * C_::A_FLOATS + 16 * (pq - C_::NA) * 16;
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + (size_t)m0 * p.lda * EB, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.W)) + (size_t)n0 * p.ldw * EB, 0, 0xffffffff, 0x00020000);
    int koff = 0;
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < C_::NI; ++u) {
            const int boff = u < UA ? buf * C_::A_FLOATS : buf * C_::B_FLOATS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(u < UA ? rsA : rsW, (lptr_t)(smem + dst[u] + boff), 16, voff[u], koff, 0, 0);
        }
        koff += 64;
    };

    // Accumulators start as bias (+ residual / + PE row).  All loads are issued back to back behind WAVE-UNIFORM branches and waited for
    // once: with per-element runtime conditions hipcc branches around every load and waits vmcnt(0) after each one -- 16-32 serialized
    // L2 round trips per tile, fully exposed at one workgroup per CU.  Rows / columns past the edge read a clamped (valid) address: their
    // accumulators are never stored.
    f32x16 acc[TM][TN];
    if constexpr (ET == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    } else {
        const bool has_bias = p.bias != nullptr;
        const bool has_ext = (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) && !bf16_late_ext(p, n0, BN);
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
                const int er = p.epilogue == MMDM_EPI_BIAS_PE ? rowc % p.period : rowc;
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

    const int nkt = p.K / (ET == 1 ? 2 * BK : BK);          // a 64-byte row step = 32 bf16 or 64 fp8
    stage(0);

    const int sw = (l31 >> 2) & 3;
    const int a_row = (wm * (32 * TM) + l31) * 16;
    const int b_row = (wn * (32 * TN) + l31) * 16;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage(cur ^ 1);
        const float* Ac = As + cur * C_::A_FLOATS + a_row;
        const float* Bc = Bs + cur * C_::B_FLOATS + b_row;
        if constexpr (ET == 1) {
            // fp8: ONE v_mfma_scale_f32_32x32x64_f8f6f4 per output tile and 64-byte row step (e4m3 on both sides, unit E8M0 scales): the
            // block-scaled form runs the 64-deep product in the time of two K = 16 fp8 MFMAs, i.e. at twice the non-scaled fp8 / bf16
            // rate (MI355X_MICROARCH.md, Matrix cores).  A lane's 32 operand bytes are its two 16-byte fragments of the step (chunks lh and
            // 2 + lh of the row); A and W use the same positions, so every product pairs the same k on both sides.
            typedef int v8i __attribute__((ext_vector_type(8)));
            typedef int v4i __attribute__((ext_vector_type(4)));
            v8i af8[TM], bf8[TN];
            const int c0 = 4 * (lh ^ sw), c1 = 4 * ((2 + lh) ^ sw);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const v4i lo = __builtin_bit_cast(v4i, *reinterpret_cast<const f32x4*>(Ac + i * 32 * 16 + c0));
                const v4i hi = __builtin_bit_cast(v4i, *reinterpret_cast<const f32x4*>(Ac + i * 32 * 16 + c1));
                af8[i] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const v4i lo = __builtin_bit_cast(v4i, *reinterpret_cast<const f32x4*>(Bc + j * 32 * 16 + c0));
                const v4i hi = __builtin_bit_cast(v4i, *reinterpret_cast<const f32x4*>(Bc + j * 32 * 16 + c1));
                bf8[j] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf8[j], af8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int cg = 4 * ((2 * kb + lh) ^ sw);
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Ac + i * 32 * 16 + cg));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Bc + j * 32 * 16 + cg));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        }
    }
    // MFMA -> VALU hazard across the loop-exit branch: see MFMA_SETTLE in attn_f32.hip
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]));

    if (!(p.tst && bf16_finish_t<TM, TN, ET, BM, BN, C_::NWAVES, C_::SMEM_BYTES>(p, acc, m0, n0, wm, wn, lane, smem)))
        bf16_finish<TM, TN, ET, BM>(p, acc, m0, n0, wm, wn, l31, lh);
#endif
}

template <int TM_, int TN_, int ET = 0>
int launch(BArgs a, hipStream_t st) {
    using C_ = BCfg<TM_, TN_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    mmdm_note_gemm("%s<%d,%d>", ET == 1 ? "gemm_fp8" : "gemm_bf16", TM_, TN_);
    hipLaunchKernelGGL((gemm_bf16_kernel<TM_, TN_, ET>), dim3(a.mt * a.nt), dim3(C_::THREADS), C_::SMEM_BYTES, st, a);
    return mmdm_check_launch(ET == 1 ? "gemm_fp8" : "gemm_bf16");
}

template <int TM_, int TN_, int ET = 0>
int set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<TM_, TN_, ET>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, BCfg<TM_, TN_>::SMEM_BYTES);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_bf16): %s", hipGetErrorString(e));
    return MMDM_OK;
}

// ---- W in fragment order, straight from global memory (the structure that took the fp32-split GEMM from 190 to 215-220 TFLOP/s) -------------
// Tile 128 x 256 of four waves side by side, each 128 x 64 (TM = 4, TN = 2: a k-block of 32 bytes per row is 4 A fragment reads from LDS and
// 2 B fragment loads from global memory for 8 MFMAs -- 16 in the fp8 form).  K step = 128 bytes per row (64 bf16 / 128 fp8 = 4 k-blocks:
// 32 / 64 MFMAs per wave between barriers instead of 8 / 16).  LDS holds A only, three stages of two 64-byte-row images; the stage after
// next and the next step's B fragments are requested behind the MFMAs of the step's first k-block.  Two workgroups per CU (registers).
// W: mmdm_pack_weight_frag -- block (32 rows, 32 bytes of k) = the 1 KiB one wave-wide 16-byte load delivers, lane (l31, lh) <- row l31, bytes 16 lh.
// Accumulators start as in gemm_bf16_kernel and k ascends the same way: bit-identical results.
constexpr int WSMEM_BYTES = 3 * 2 * 128 * 16 * 4;      // operand stages of gemm_bf16w_kernel: three A stages of two 64-byte-row images (48 KB)
constexpr int wsmem_total(int ET, int TN) { return WSMEM_BYTES + (ET == 1 ? (128 + 2 * 128 * TN) * 4 : 0); }      // fp8: + [a_scale | w_scale | bias] of the tile
template <int ET, int TN = 2, bool TL = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16w_kernel(BArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long t_entry = 0, t_r0 = 0, t_c0 = 0, t_r1 = 0, t_c1 = 0;
    if constexpr (TL) t_entry = __builtin_amdgcn_s_memrealtime();
    constexpr int TM = 4, BM = 128, BN = 128 * TN, NW = 4, NKB = 4;      // TN = 1: 128 x 128 tiles for the N <= 1024 GEMMs (600 tiles of 128 x 256 leave half of the second round empty)
    constexpr int HALF = BM * 16, STAGE = 2 * HALF;          // 4-byte units: one 64-byte-row image, one stage
    constexpr int NIA = 2 * (BM / 16) / NW;                 // LDS-DMA pieces per wave and step (4)
    constexpr int NLB = NKB * TN;                            // B fragment loads per wave and step (8)
    constexpr int EB = ET == 1 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int mi, ni;
    {
        const int GM = 8, per_g = GM * p.nt, g = swz / per_g, rem = swz - g * per_g;
        const int gm = min(GM, p.mt - g * GM);
        ni = rem / gm; mi = g * GM + rem - ni * gm;
    }
    const int m0 = mi * BM, n0 = ni * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = 0, wn = wave;
    const int l31 = lane & 31, lh = lane >> 5;
    const int kbytes = p.K * EB;
    const int nkt = kbytes / 128;

    int voff[NIA], dst[NIA];
#pragma unroll
    for (int u = 0; u < NIA; ++u) {
        const int pq = wave + NW * u;                        // 0 .. 15: image (pq / 8), 16-row group (pq % 8)
        const int half = pq >> 3, pp = pq & 7;
        const int prow = lane >> 2, pc = lane & 3;
        const int trow = 16 * pp + prow;
        const int gch = pc ^ ((trow >> 2) & 3);
        voff[u] = min(trow, p.M - 1 - m0) * p.lda * EB + 64 * half + 16 * gch;
        dst[u] = half * HALF + 16 * pp * 16;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.A)) + (size_t)m0 * p.lda * EB, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.W)), 0, 0xffffffff, 0x00020000);
    auto stage = [&](int buf, int kt) {                     // kt clamped: the loop stays branch-free
        const int koff = min(kt, nkt - 1) * 128;
#pragma unroll
        for (int u = 0; u < NIA; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + dst[u] + buf * STAGE), 16, voff[u], koff, 0, 0);
    };
    int voffW[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) voffW[j] = ((n0 >> 5) + wn * TN + j) * (kbytes >> 5) * 1024 + lane * 16;
    auto ldb = [&](int kt, bf16x8 (&b)[NKB][TN]) {
        const int so = min(kt, nkt - 1) * (NKB * 1024);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kb][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW[j] + kb * 1024, so, 0));
    };
    // half a step of W fragments (k-blocks 2 half, 2 half + 1): the fp8 128 x 256 form keeps two HALF sets instead of two whole ones
    auto ldbh = [&](int kt, int half, bf16x8 (&b)[2][TN]) {
        const int so = min(kt, nkt - 1) * (NKB * 1024);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kb][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW[j] + (2 * half + kb) * 1024, so, 0));
    };


    f32x16 acc[TM][TN];
    float* const sc = smem + WSMEM_BYTES / 4;               // fp8: [BM] a_scale | [BN] w_scale | [BN] bias of this tile (read by the epilogue)
    static_assert(BN <= 256, "one w_scale / bias element per thread");
    float sc_a = 0.f, sc_w = 1.f, sc_b = 0.f;
    if constexpr (ET == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // the epilogue's per-row / per-column operands (128 + 2 BN floats; thread t takes element t of each) are requested now, ahead
        // of the first operand stages, and parked in LDS once those requests are out (below): their latency passes under the pipeline's own
        {
            const int rowc = min(m0 + (tid & 127), p.M - 1);
            sc_a = p.a_const;
            if (p.a_scale) sc_a = p.a_scale[rowc];
            const int col = min(n0 + tid, p.N - 1);
            if (p.w_scale) sc_w = p.w_scale[col];
            if (p.bias) sc_b = p.bias[col];
        }
    } else {
        const bool has_bias = p.bias != nullptr;
        const bool has_ext = (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) && !bf16_late_ext(p, n0, BN);
        int colc[TN][4];
        f32x4 bv[TN][4];
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
        for (int i = 0; i < TM; ++i) {
            const int rowc = min(m0 + i * 32 + l31, p.M - 1);
            const int er = p.epilogue == MMDM_EPI_BIAS_PE ? rowc % p.period : rowc;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    f32x4 v = bv[j][qd];
                    if (has_ext) v += *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + colc[j][qd]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] = v[c];
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the pipeline below counts its own requests only
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
    }

    const int sw = (l31 >> 2) & 3;
    const int a_row = l31 * 16;
    bf16x8 fa0[TM], fa1[TM];
    bf16x8 bx[NKB][TN], by[NKB][TN];
    auto rda = [&](int buf, int kb, bf16x8 (&af)[TM]) {
        const float* Ac = smem + buf * STAGE + (kb >> 1) * HALF + a_row + 4 * ((2 * (kb & 1) + lh) ^ sw);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(Ac + i * 32 * 16));
    };
    auto mm = [&](const bf16x8 (&af)[TM], const bf16x8 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    };
    // fp8: the block-scaled 64-deep MFMA takes the fragments of TWO k-blocks (this lane's 16 bytes of each) per operand, exactly as the
    // staged kernel pairs the two fragments of its 64-byte row step -- same products, same order: the two kernels stay bit-identical.
    typedef int v8i __attribute__((ext_vector_type(8)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto cat = [](const bf16x8& lo, const bf16x8& hi) {
        const v4i a = __builtin_bit_cast(v4i, lo), b = __builtin_bit_cast(v4i, hi);
        return v8i{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };
    auto mm64 = [&](const bf16x8 (&al)[TM], const bf16x8 (&ah)[TM], const bf16x8 (&bl)[TN], const bf16x8 (&bh)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat(bl[j], bh[j]), cat(al[i], ah[i]), acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    };
    constexpr int NM = TM * TN;                              // MFMAs per k-block (bf16) / per pair of k-blocks (fp8)
    constexpr int NV = NLB + NIA;                            // VMEM instructions of a step, all behind the first k-block's MFMAs
    // fp8 step: f0 holds the fragments of k-block 0 on entry and f2 receives those of the next step's k-block 0
    auto step8 = [&](int kt, int cur, int nxt, int nn, bf16x8 (&b)[NKB][TN], bf16x8 (&bn)[NKB][TN], bf16x8 (&f0)[TM], bf16x8 (&f1)[TM], bf16x8 (&f2)[TM]) {
        rda(cur, 1, f1);
        ldb(kt + 1, bn);
        stage(nn, kt + 2);
        mm64(f0, f1, b[0], b[1]);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int u = 0; u < NM - 2; ++u) { __builtin_amdgcn_sched_group_barrier(0x010, (NV + NM - 3) / (NM - 2), 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x010, NV, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        rda(cur, 2, f0);
        rda(cur, 3, f1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");      // A stage kt+1 landed (requested during step kt-1); this step's requests may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // every wave has read all of stage kt and sees stage kt+1
        __builtin_amdgcn_sched_barrier(0);
        rda(nxt, 0, f2);
        mm64(f0, f1, b[2], b[3]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // fp8, 128 x 256 tiles: 128 accumulators + three A fragment sets leave room for ONE step of W fragments, not two (25 spills with the
    // whole-step double buffer).  The W fragments are therefore requested half a step ahead: k-blocks 2, 3 of this step behind the first
    // 64-deep MFMA group (which runs on k-blocks 0, 1), k-blocks 0, 1 of the next step behind the second.  Same products in the same order.
    auto step8h = [&](int kt, int cur, int nxt, int nn, bf16x8 (&bl)[2][TN], bf16x8 (&bh)[2][TN], bf16x8 (&f0)[TM], bf16x8 (&f1)[TM], bf16x8 (&f2)[TM]) {
        constexpr int NVH = NLB / 2 + NIA;
        rda(cur, 1, f1);
        ldbh(kt, 1, bh);
        stage(nn, kt + 2);
        mm64(f0, f1, bl[0], bl[1]);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int u = 0; u < NM - 2; ++u) { __builtin_amdgcn_sched_group_barrier(0x010, (NVH + NM - 3) / (NM - 2), 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x010, NVH, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        rda(cur, 2, f0);
        rda(cur, 3, f1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NVH) : "memory");     // A stage kt+1 landed (requested during step kt-1); this step's requests may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rda(nxt, 0, f2);
        mm64(f0, f1, bh[0], bh[1]);
        ldbh(kt + 1, 0, bl);                                             // (bl was consumed by the first group)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 1);
#pragma unroll
        for (int u = 0; u < NLB / 2; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x010, 1, 1); }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1 - NLB / 2, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int kt, int cur, int nxt, int nn, bf16x8 (&b)[NKB][TN], bf16x8 (&bn)[NKB][TN]) {
        rda(cur, 1, fa1);
        ldb(kt + 1, bn);
        stage(nn, kt + 2);
        mm(fa0, b[0]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
        for (int u = 0; u < NM - 2; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, (NV + NM - 3) / (NM - 2), 0); }
        __builtin_amdgcn_sched_group_barrier(0x010, NV, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        rda(cur, 2, fa0);
        mm(fa1, b[1]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        rda(cur, 3, fa1);
        mm(fa0, b[2]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");      // A stage kt+1 landed (requested during step kt-1); this step's requests may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // every wave has read all of stage kt and sees stage kt+1
        __builtin_amdgcn_sched_barrier(0);
        rda(nxt, 0, fa0);
        mm(fa1, b[3]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
        __builtin_amdgcn_sched_group_barrier(0x100, TM, 3);
        __builtin_amdgcn_sched_group_barrier(0x008, NM - 1, 3);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (TL) { t_r0 = __builtin_amdgcn_s_memrealtime(); t_c0 = __builtin_readcyclecounter(); }
    constexpr bool HALFB = ET == 1 && TN == 2;
    stage(0, 0);
    bf16x8 bl[2][TN], bh[2][TN];                           // HALFB: the two half sets of W fragments (bx / by unused)
    if constexpr (HALFB) ldbh(0, 0, bl);
    else ldb(0, bx);
    stage(1, 1);
    if constexpr (ET == 1) {                               // (older than every request above: the wait the compiler puts here leaves those in flight)
        if (tid < 128) sc[tid] = sc_a;
        if (tid < BN) { sc[BM + tid] = sc_w; sc[BM + BN + tid] = sc_b; }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIA) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rda(0, 0, fa0);
    int cur = 0;
    bf16x8 fa2[TM];
#ifndef MMDM_W_EXP
#define MMDM_W_EXP 0            // experiments of tools/ (variant builds, never the shipped library): 1 = no epilogue, 2 = no K loop
#endif
    for (int kt = 0; kt < (MMDM_W_EXP == 2 ? 0 : nkt); kt += 2) {                  // nkt is even (host check): the two B register sets swap roles every step
        const int c1 = cur == 2 ? 0 : cur + 1, c2 = c1 == 2 ? 0 : c1 + 1;
        if constexpr (HALFB) {
            step8h(kt, cur, c1, c2, bl, bh, fa0, fa1, fa2);
            step8h(kt + 1, c1, c2, cur, bl, bh, fa2, fa1, fa0);
        } else if constexpr (ET == 1) {
            step8(kt, cur, c1, c2, bx, by, fa0, fa1, fa2);
            step8(kt + 1, c1, c2, cur, by, bx, fa2, fa1, fa0);
        } else {
            step(kt, cur, c1, c2, bx, by);
            step(kt + 1, c1, c2, cur, by, bx);
        }
        cur = c2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the surplus requests of the last steps
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]));
    if constexpr (TL) { t_r1 = __builtin_amdgcn_s_memrealtime(); t_c1 = __builtin_readcyclecounter(); }
#if MMDM_W_EXP == 1
    {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
        if (sacc == 1234.5678f) static_cast<float*>(p.C)[tid] = sacc;
    }
#else
    if (!(p.tst && bf16_finish_t<TM, TN, ET, BM, BN, NW, WSMEM_BYTES>(p, acc, m0, n0, wm, wn, lane, smem, ET == 1 ? sc : nullptr)))
        bf16_finish<TM, TN, ET, BM>(p, acc, m0, n0, wm, wn, l31, lh);
#endif
    if constexpr (TL) {
        const unsigned long long t_iss = __builtin_amdgcn_s_memrealtime();      // every instruction of the epilogue issued; its stores may be in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.tl && tid == 0) {
            unsigned long long* o = p.tl + 4 * (size_t)blockIdx.x;
            o[0] = t_c1 - t_c0; o[1] = (t_r1 - t_r0) | ((t_iss - t_r1) << 32); o[2] = __builtin_amdgcn_s_memrealtime() - t_entry;
            o[3] = (unsigned long long)nkt | ((t_r0 - t_entry) << 32);      // upper half: ticks from kernel entry to the loop (accumulator start + first requests)
        }
    }
#endif
}

template <int ET, int TN = 2>
int launch_w(BArgs a, hipStream_t st) {
    a.mt = (a.M + 127) / 128;
    a.nt = a.N / (128 * TN);
    mmdm_note_gemm("%s<14,4%d>", ET == 1 ? "gemm_fp8w" : "gemm_bf16w", TN);
    if (a.tl) hipLaunchKernelGGL((gemm_bf16w_kernel<ET, TN, true>), dim3(a.mt * a.nt), dim3(256), wsmem_total(ET, TN) + g_bf16_lds_pad, st, a);
    else hipLaunchKernelGGL((gemm_bf16w_kernel<ET, TN>), dim3(a.mt * a.nt), dim3(256), wsmem_total(ET, TN) + g_bf16_lds_pad, st, a);
    return mmdm_check_launch(ET == 1 ? "gemm_fp8w" : "gemm_bf16w");
}

// rows of `row_bytes` operand bytes [N][ld_bytes] -> fragment order: block (32 rows, 32 bytes) = 1 KiB, chunk (lh, l31) <- row l31, bytes 16 lh
__global__ void pack_frag_kernel(const unsigned char* __restrict__ in, size_t ld_bytes, unsigned char* __restrict__ out, int N, int row_bytes) {
    const size_t chunks = (size_t)N * row_bytes / 16;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < chunks; o += (size_t)gridDim.x * blockDim.x) {
        const int ln = (int)(o & 63), l31 = ln & 31, lh = ln >> 5;
        const size_t blk = o >> 6;
        const int kb = (int)(blk % (size_t)(row_bytes >> 5)), nb = (int)(blk / (size_t)(row_bytes >> 5));
        *reinterpret_cast<f32x4*>(out + o * 16) = *reinterpret_cast<const f32x4*>(in + (size_t)(nb * 32 + l31) * ld_bytes + kb * 32 + lh * 16);
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (__bf16)in[i];
}

// Row-wise e4m3 quantisation: scale[r] = max|x[r,:]| / 448 (1 for an all-zero row), q[r,k] = e4m3(x[r,k] / scale[r]).  One wave per row.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const float* __restrict__ in, int ld_in, unsigned char* __restrict__ out, int ld_out,
                                                              float* __restrict__ scale, int rows, int K) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = in + (size_t)row * ld_in;
    float m = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + k);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float sc = m > 0.f ? m * (1.0f / 448.0f) : 1.0f, inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    for (int k = 4 * lane; k < K; k += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + k) * inv;
        *reinterpret_cast<unsigned*>(out + (size_t)row * ld_out + k) = pack_fp8x4(v);
    }
}

int g_bf16_cfg = -1;
unsigned long long* g_bf16_tl = nullptr;

}  // namespace

int mmdm_gemm_bf16_init(void) {
    int rc;
    if ((rc = mmdm_fp8p_init())) return rc;
    if ((rc = set_attr<22, 22>())) return rc;
    if ((rc = set_attr<42, 22>())) return rc;
    if ((rc = set_attr<42, 42>())) return rc;
    if ((rc = set_attr<22, 21>())) return rc;
    if ((rc = set_attr<22, 22, 1>())) return rc;
    if ((rc = set_attr<42, 22, 1>())) return rc;
    if ((rc = set_attr<42, 42, 1>())) return rc;
    if ((rc = set_attr<22, 21, 1>())) return rc;
    for (const void* f : {reinterpret_cast<const void*>(&gemm_bf16w_kernel<0>),
                          reinterpret_cast<const void*>(&gemm_bf16w_kernel<0, 1>), reinterpret_cast<const void*>(&gemm_bf16w_kernel<1, 1>),
                          reinterpret_cast<const void*>(&gemm_bf16w_kernel<1, 2>), reinterpret_cast<const void*>(&gemm_bf16w_kernel<1, 2, true>),
                          reinterpret_cast<const void*>(&gemm_bf16w_kernel<0, 2, true>),
                          reinterpret_cast<const void*>(&gemm_bf16w_kernel<0, 1, true>), reinterpret_cast<const void*>(&gemm_bf16w_kernel<1, 1, true>)}) {
        hipError_t e2 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);      // (the operand stages need wsmem_total(); "bf16_lds_pad" may ask for more)
        if (e2 != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_bf16w): %s", hipGetErrorString(e2));
    }
    return MMDM_OK;
}

// diagnostics of this translation unit (mmdm_diag_set): tile override; timeline buffer (4 u64 per workgroup of the next packed launches)
bool mmdm_diag_gemm_bf16(const char* key, long long v) {
    if (!strcmp(key, "bf16_cfg")) g_bf16_cfg = (int)v;
    else if (!strcmp(key, "bf16_tst")) g_bf16_tst = (int)v;
    else if (!strcmp(key, "bf16_lds_pad")) g_bf16_lds_pad = (int)v;
    else if (!strcmp(key, "fp8p")) g_fp8p = (int)v;
    else if (!strncmp(key, "fp8p_", 5)) return mmdm_diag_gemm_fp8p(key, v);
    else if (!strcmp(key, "bf16_timeline")) g_bf16_tl = reinterpret_cast<unsigned long long*>((uintptr_t)v);
    else return false;
    return true;
}

extern "C" int mmdm_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    if (n <= 0) return MMDM_OK;
    if (!in || !out) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_f32_to_bf16: null argument");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), in, static_cast<__bf16*>(out), (size_t)n);
    return mmdm_check_launch("f32_to_bf16");
}

// W [N][row_bytes] (bf16: row_bytes = 2 K; fp8: K) -> fragment order for the *_packed entry points; N % 32 == 0, row_bytes % 32 == 0
extern "C" int mmdm_pack_weight_frag(const void* W, int64_t ld_bytes, void* out, int N, int row_bytes, void* stream) {
    if (N == 0) return MMDM_OK;
    if (!W || !out || N < 0 || row_bytes <= 0 || (N & 31) || (row_bytes & 31) || ld_bytes < row_bytes || (ld_bytes & 15) ||
        (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_pack_weight_frag: needs N %% 32 == 0, row bytes %% 32 == 0 and 16-byte aligned rows (N=%d row_bytes=%d)", N, row_bytes);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned char*>(W), (size_t)ld_bytes,
                       static_cast<unsigned char*>(out), N, row_bytes);
    return mmdm_check_launch("pack_weight_frag");
}

extern "C" int mmdm_linear_bf16_packed(const void* A, int lda, const void* W_packed, const float* bias, void* C, int ldc, int out_bf16,
                                       int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_bf16_ex(A, lda, W_packed, 0, bias, C, ldc, out_bf16, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, stream);
}

extern "C" int mmdm_linear_fp8_packed(const void* A, int lda, const float* a_scale, const void* W_packed, const float* w_scale, const float* bias, void* C, int ldc,
                                      int out_mode, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_fp8_ex(A, lda, a_scale, W_packed, 0, w_scale, bias, C, ldc, out_mode, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, 1.0f, 1.0f, stream);
}

extern "C" int mmdm_linear_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int out_bf16,
                                int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_bf16_ex(A, lda, W, ldw, bias, C, ldc, out_bf16, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, stream);
}

int mmdm_linear_bf16_ex(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int out_bf16,
                        int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* bf16_copy, int ld2, int copy_cols, void* stream) {
    mmdm_note_gemm_reset();
    if (M == 0 || N == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    const bool packed = ldw == 0;                 // W in fragment order (mmdm_pack_weight_frag): gemm_bf16w_kernel
    if (packed) {
        if ((N & 127) || (K & 127)) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_bf16_packed: needs N %% 128 == 0 and K %% 128 == 0 (N=%d K=%d)", N, K);
        ldw = K;
    }
    if (!A || !W || !C || M < 0 || N < 0 || K <= 0 || lda < K || ldw < K || ldc < N)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_bf16: bad shape M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
    if (epilogue < MMDM_EPI_BIAS || epilogue > MMDM_EPI_BIAS_SILU) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_bf16: unknown epilogue %d", epilogue);
    const bool ext = epilogue == MMDM_EPI_BIAS_RESID || epilogue == MMDM_EPI_BIAS_PE;
    if (ext && (!extra || ld_extra < N)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_bf16: epilogue %d needs `extra` with ld >= N", epilogue);
    if (epilogue == MMDM_EPI_BIAS_PE && period <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_bf16: PE epilogue needs period > 0");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((K & 31) || (lda & 7) || (ldw & 7) || !al16(A) || !al16(W))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_bf16: needs K %% 32 == 0 and 16-byte aligned bf16 rows");
    if ((N & 3) || (ldc & 3) || !al16(C) || (bias && !al16(bias)) || (ext && ((ld_extra & 3) || !al16(extra))))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_bf16: needs N %% 4 == 0 and 16-byte aligned output / bias / residual rows");
    BArgs a;
    a.A = static_cast<const __bf16*>(A); a.W = static_cast<const __bf16*>(W); a.bias = bias; a.C = C; a.extra = extra;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ld_extra = ld_extra;
    a.M = M; a.N = N; a.K = K; a.epilogue = epilogue; a.period = period > 0 ? period : 1; a.out_bf16 = out_bf16;
    a.mt = a.nt = 0;
    a.a_scale = a.w_scale = nullptr; a.a_const = a.out_scale = 1.f; a.tl = packed ? g_bf16_tl : nullptr;
    a.P2 = static_cast<__bf16*>(bf16_copy); a.p2_cols = copy_cols; a.ld2 = ld2;
    a.tst = g_bf16_tst;
    if (bf16_copy && ((ld2 & 3) || (copy_cols & 3) || !al16(bf16_copy))) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_bf16: second output needs 8-byte aligned bf16 rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (packed) {
        // 128 x 256 tiles halve the A-fragment LDS reads per MFMA, but N <= 1024 gives them too few tiles at M = 19 200 (600 = 1.17 rounds of the
        // 512 slots); from two rounds up they win (B = 32 / 64 shards: bf16 25.5 -> 24.8 / 50.0 -> 47.9 ms/step, bf16_fp8 21.5 -> 21.3 / 42.0 -> 41.0)
        const bool few = (long)((M + 127) / 128) * (N / 256) < 1024;
        const bool narrow = (N & 255) || (g_bf16_cfg == 11) || (g_bf16_cfg != 12 && N <= 1024 && few);
        return narrow ? launch_w<0, 1>(a, st) : launch_w<0, 2>(a, st);
    }
    switch (g_bf16_cfg) {
        case 0: return launch<22, 22>(a, st);
        case 1: return launch<42, 22>(a, st);
        case 2: return launch<42, 42>(a, st);
        case 3: return launch<22, 21>(a, st);
        default: return launch<42, 22>(a, st);
    }
}

// ---------------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) operands: BASELINE configs[4] "fp8 MFMA QKV/FFN GEMMs"
// ---------------------------------------------------------------------------------------------------------
extern "C" int mmdm_quantize_rows_fp8(const float* in, int ld_in, void* out, int ld_out, float* scale, int rows, int K, void* stream) {
    if (rows == 0) return MMDM_OK;
    if (!in || !out || !scale || rows < 0 || K <= 0 || (K & 3) || (ld_in & 3) || (ld_out & 3) || ld_in < K || ld_out < K ||
        (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 3))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_quantize_rows_fp8: bad arguments (K and the row strides must be multiples of 4, rows 16-byte aligned)");
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in, ld_in,
                       static_cast<unsigned char*>(out), ld_out, scale, rows, K);
    return mmdm_check_launch("quant_rows_fp8");
}

extern "C" int mmdm_linear_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* C, int ldc,
                               int out_mode, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_fp8_ex(A, lda, a_scale, W, ldw, w_scale, bias, C, ldc, out_mode, M, N, K, epilogue, extra, ld_extra, period, nullptr, 0, 0, 1.0f, 1.0f, stream);
}

// a_const: de-quantisation factor of A when a_scale is null; out_scale: an fp8 output is e4m3(value * out_scale).  The handle uses the pair
// for the GELU tensor between the two FFN GEMMs, whose producer sees a 64 x 64 piece of a row and cannot take a row maximum: a static power
// of two (x 16 stored, 1/16 applied by the consumer) moves e4m3's normal range from |v| in [2^-6, 448] to [2^-10, 28].
int mmdm_linear_fp8_ex(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* C, int ldc,
                       int out_mode, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* bf16_copy, int ld2, int copy_cols,
                       float a_const, float out_scale, void* stream) {
    mmdm_note_gemm_reset();
    if (M == 0 || N == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    const bool packed = ldw == 0;
    if (packed) {
        if ((N & 127) || (K & 255)) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_fp8_packed: needs N %% 128 == 0 and K %% 256 == 0 (N=%d K=%d)", N, K);
        ldw = K;
    }
    if (!A || !W || !C || M < 0 || N < 0 || K <= 0 || lda < K || ldw < K || ldc < N)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_fp8: bad shape M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
    if (out_mode < 0 || out_mode > 2) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_fp8: out_mode must be 0 (fp32), 1 (bf16) or 2 (fp8 at unit scale)");
    if (epilogue < MMDM_EPI_BIAS || epilogue > MMDM_EPI_BIAS_SILU) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_fp8: unknown epilogue %d", epilogue);
    const bool ext = epilogue == MMDM_EPI_BIAS_RESID || epilogue == MMDM_EPI_BIAS_PE;
    if (ext && (!extra || ld_extra < N)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_fp8: epilogue %d needs `extra` with ld >= N", epilogue);
    if (epilogue == MMDM_EPI_BIAS_PE && period <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_fp8: PE epilogue needs period > 0");
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((K & 63) || (lda & 15) || (ldw & 15) || !al16(A) || !al16(W))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_fp8: needs K %% 64 == 0 and 16-byte aligned fp8 rows");
    if ((N & 3) || (ldc & 3) || !al16(C) || (bias && !al16(bias)) || (w_scale && !al16(w_scale)) || (ext && ((ld_extra & 3) || !al16(extra))))
        return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_fp8: needs N %% 4 == 0 and 16-byte aligned output / bias / scale / residual rows");
    BArgs a;
    a.A = static_cast<const __bf16*>(A); a.W = static_cast<const __bf16*>(W); a.bias = bias; a.C = C; a.extra = extra;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ld_extra = ld_extra;
    a.M = M; a.N = N; a.K = K; a.epilogue = epilogue; a.period = period > 0 ? period : 1; a.out_bf16 = out_mode;
    a.mt = a.nt = 0;
    a.a_scale = a_scale; a.w_scale = w_scale;
    a.a_const = a_const; a.out_scale = out_scale; a.tl = packed ? g_bf16_tl : nullptr;
    a.P2 = static_cast<__bf16*>(bf16_copy); a.p2_cols = copy_cols; a.ld2 = ld2;
    a.tst = g_bf16_tst;
    if (bf16_copy && ((ld2 & 3) || (copy_cols & 3) || !al16(bf16_copy))) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_linear_fp8: second output needs 8-byte aligned bf16 rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (packed && g_fp8p && !bf16_copy && a.tst && !a.tl) {
        // the persistent form: a tile's epilogue under the next tile's K loop (gemm_fp8p.hip; bit-identical results)
        Fp8pArgs q;
        q.A = A; q.W = W; q.a_scale = a_scale; q.w_scale = w_scale; q.bias = bias; q.C = C; q.extra = extra;
        q.lda = lda; q.ldc = ldc; q.ld_extra = ld_extra; q.M = M; q.N = N; q.K = K; q.epilogue = epilogue; q.period = a.period; q.out_mode = out_mode;
        q.a_const = a_const; q.out_scale = out_scale; q.mt = q.nt = q.ntiles = 0; q.tl = nullptr;
        if (mmdm_fp8p_covers(q) && (g_fp8p >= 2 || (out_mode == 1 && N <= 2048 && epilogue == MMDM_EPI_BIAS))) return mmdm_fp8p_launch(q, st);
    }
    if (packed) {
        // fp8 packed: 128 x 256 tiles (W fragments requested half a step ahead: gemm_bf16w_kernel, HALFB) for the large shards only -- measured
        // bf16_fp8 B = 64: 41.7 -> 41.0 ms/step, B = 16: 11.07 vs 11.08 (the shorter prefetch distance costs what the halved LDS reads
        // save) -- 128 x 128 otherwise (MMDM_BF16_CFG=12 / 13 force the wide / the narrow form)
        // (round 6: for the GELU epilogue the wide tile is 5.5 % faster stand-alone at M = 19 200 -- 63.1 -> 59.6 us -- and SLOWER inside the two-stream step:
        //  9.183 vs 9.14 ms/step, tools/fp8_step_ab.py on one box; the rule stays "large shards only")
        const bool wide = (N & 255) == 0 && g_bf16_cfg != 11 && g_bf16_cfg != 13 && (g_bf16_cfg == 12 || (long)((M + 127) / 128) * (N / 256) >= 2400);
        return wide ? launch_w<1, 2>(a, st) : launch_w<1, 1>(a, st);
    }
    switch (g_bf16_cfg) {
        case 0: return launch<22, 22, 1>(a, st);
        case 2: return launch<42, 42, 1>(a, st);
        case 3: return launch<22, 21, 1>(a, st);
        default: return launch<42, 22, 1>(a, st);
    }
}
