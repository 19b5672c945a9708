// Persistent packed fp8 GEMM for gfx950 (BASELINE configs[4]; round 6): C = dequant(A W^T) + b with tile i's epilogue issued UNDER tile i + 1's K loop.
//
// Why (LAB_NOTES.md round 5 / 6): at K = 1024 a 128 x 128 tile's K loop is 8 steps of 8 block-scaled 64-deep MFMAs per wave
// (4096 matrix-pipe cycles) and its de-quantising epilogue is as long again in wave time -- a workgroup of gemm_bf16w_kernel<1, 1> spends
// half of its life outside the loop, and three such workgroups per CU overlap only statistically: the matrix pipe is 28-32 % busy.  Here a
// workgroup is PERSISTENT -- it walks a fixed sequence of tiles -- and every wave keeps TWO accumulator sets: while the MFMAs of tile i + 1
// accumulate into one, the finished tile i leaves the other through four phases spread over the first eight K steps of tile i + 1:
//     step 2p     (first half)   W(p): row tile p of tile i: de-quantise, bias, activation, convert -> 32-row image in LDS   (VALU + ds_write)
//     step 2p + 1 (first half)   R(p): image -> C as whole rows (ds_read_b128 + buffer_store_b128), residual rows added here   (LDS + VMEM)
// separated by the K loop's own barrier (one per step), so the epilogue costs no barrier of its own, no prologue latency (the next tile's
// first A stages and W fragments are requested by the last two steps of the current one: the request stream never drains) and its VALU
// work issues between MFMAs of the same wave.  128 accumulator registers: two workgroups per CU (256 registers per wave).
//
// Same arithmetic as gemm_bf16w_kernel<1, .>: the same products in the same k order, the same de-quantisation expression, activation and
// conversion -- results are bit-identical (tests/test_gpu_fp8.py compares the two kernels bitwise).
//
// Tile walk: XCD x (blockIdx & 7) owns a contiguous eighth of the tile sequence (its rows of A and columns of W stay in one L2), its G / 8
// workgroups take that range round-robin -- static, no atomics: a tile counter would have to return inside the K loop, where vector memory
// operations retire in order.
#include <hip/hip_runtime.h>
#include <string.h>
#include <type_traits>
#include "kernels.h"
#include "gemm_fp8p.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 128, BN = 128, NW = 4, TM = 4, NKB = 4;
constexpr int HALF = BM * 16, STAGE = 2 * HALF;        // 4-byte units: one 64-byte-row image of the tile's 128 rows; one stage = 128 operand bytes per row
#ifndef FP8P_DEEP
#define FP8P_DEEP 0             // 1: the deeper request pipeline (measured, not faster: see step()); variant builds of tools/ only
#endif
constexpr bool DEEP = FP8P_DEEP != 0;
constexpr int NSTG = DEEP ? 4 : 3;                     // A stages in LDS: the stage of step kt + NSTG - 1 is requested during step kt
constexpr int NIA = 2 * (BM / 16) / NW;                // LDS-DMA pieces per wave and step (4)
constexpr int NLB = NKB;                               // W fragment loads per wave and step (4)
constexpr int NV = NIA + NLB;                          // operand requests of a step
constexpr int IMG_BYTES = 32 * BN * 2;                 // the epilogue's image: 32 rows of the widest output form covered (bf16)
constexpr int SC_FLOATS = 3 * 128;                     // per tile: a_scale of its rows | w_scale | bias of its columns
#ifdef FP8P_TL
constexpr int TL_BYTES = (32 + 256) * 8;               // the stamps of the diagnostic build: 32 slots and a trash row
#else
constexpr int TL_BYTES = 0;
#endif
constexpr int SMEM_BYTES = NSTG * STAGE * 4 + IMG_BYTES + 2 * SC_FLOATS * 4 + TL_BYTES;

__device__ __forceinline__ unsigned pack_fp8x4(f32x4 v) {      // OCP e4m3, RNE, saturating at +-448 (as in gemm_bf16.hip)
    float c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = fminf(fmaxf(v[i], -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
    return (unsigned)w;
}

template <int N> using IC = std::integral_constant<int, N>;

// OUT: 0 fp32, 1 bf16, 2 fp8; ACT: MMDM_EPI_BIAS / _GELU / _SILU; EXT: residual / PE rows added in R (fp32 output); NG: groups of 8 K steps (K = 1024 NG)
template <int OUT, int ACT, bool EXT, int NG>
__global__ __launch_bounds__(256, 2) void gemm_fp8p_kernel(Fp8pArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int EBO = OUT == 0 ? 4 : (OUT == 1 ? 2 : 1);        // bytes per output element
    constexpr int RB = BN * EBO, CPR = RB / 16;                   // image row: bytes, 16-byte chunks
    constexpr int LPR = CPR, RPI = 64 / LPR;                      // R: lanes per row, rows per store instruction
    constexpr int NR = 8 / RPI;                                   // R: store instructions per wave and phase (a wave stores 8 of the phase's 32 rows)
    constexpr int SWM = CPR - 1 < 15 ? CPR - 1 : 15;
    constexpr int NS = NR;                                        // the stores of an R chunk: what the next step's counted wait may leave in flight
    static_assert(8 % RPI == 0 && 32 * RB <= IMG_BYTES, "image geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const img = reinterpret_cast<char*>(smem + NSTG * STAGE);
    float* const scb = smem + NSTG * STAGE + IMG_BYTES / 4;       // [2][a_scale 128 | w_scale 128 | bias 128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int kbytes = p.K;                                       // fp8: one byte per element
    constexpr int nkt = 8 * NG;

    // ---- this workgroup's tile sequence -------------------------------------------------------------------------------------------------
    const int G8 = gridDim.x >> 3;                                // workgroups per XCD (gridDim.x is a multiple of 8)
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3;
    const int q = p.ntiles >> 3, r = p.ntiles & 7;
    const int c0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, clen = q + (xcd < r ? 1 : 0);
    if (wl >= clen) return;
    auto tile_mn = [&](int j, int& m0, int& n0) {                 // j-th tile of this workgroup -> its first row / column (gemm_bf16w_kernel's grouping: 8 row tiles x all columns)
        const int swz = c0 + min(wl + j * G8, clen - 1);
        const int GM = 8, per_g = GM * p.nt, g = swz / per_g, rem = swz - g * per_g;
        const int gm = min(GM, p.mt - g * GM);
        const int ni = rem / gm, mi = g * GM + rem - ni * gm;
        m0 = mi * BM; n0 = ni * BN;
    };
    const int ntl = (clen - wl + G8 - 1) / G8;                    // tiles of this workgroup (>= 1)

    // ---- operand addressing (as gemm_bf16w_kernel) ---------------------------------------------------------------------------------------
    int voffA[NIA], dstA[NIA];
#pragma unroll
    for (int u = 0; u < NIA; ++u) {
        const int pq = wave + NW * u;
        const int half = pq >> 3, pp = pq & 7;
        const int prow = lane >> 2, pc = lane & 3;
        const int trow = 16 * pp + prow;
        const int gch = pc ^ ((trow >> 2) & 3);
        voffA[u] = trow * p.lda + 64 * half + 16 * gch;           // (M % 128 == 0: no row clamp)
        dstA[u] = half * HALF + 16 * pp * 16;
    }
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.W)), 0, 0xffffffff, 0x00020000);
    auto rsA_of = [&](int m0) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(p.A)) + (size_t)m0 * p.lda, 0, 0xffffffff, 0x00020000); };
    auto voffW_of = [&](int n0) { return ((n0 >> 5) + wave) * (kbytes >> 5) * 1024 + lane * 16; };
    // half a step of W fragments: k-blocks 2 h, 2 h + 1 of step kt -> b[2 h], b[2 h + 1]
    auto ldbh = [&](int vw, int kt, int h, bf16x8 (&b)[NKB]) {
        const int so = kt * (NKB * 1024);
#pragma unroll
        for (int kb = 2 * h; kb < 2 * h + 2; ++kb) b[kb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, vw + kb * 1024, so, 0));
    };
    // half a stage of A: pieces 2 h, 2 h + 1 of this wave
    auto stageh = [&](int buf, __amdgpu_buffer_rsrc_t rs, int kt, int h) {
        const int koff = kt * 128;
#pragma unroll
        for (int u = 2 * h; u < 2 * h + 2; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + dstA[u] + buf * STAGE), 16, voffA[u], koff, 0, 0);
    };
    const int sw = (l31 >> 2) & 3;
    const int a_row = l31 * 16;
    // A fragment of k-block kb (0 .. 3) of stage `buf`, row tile i: this lane's 16 operand bytes
    auto rda1 = [&](int buf, int kb, int i) {
        const float* Ac = smem + buf * STAGE + (kb >> 1) * HALF + a_row + 4 * ((2 * (kb & 1) + lh) ^ sw);
        return __builtin_bit_cast(v4i, *reinterpret_cast<const f32x4*>(Ac + i * 32 * 16));
    };
    // the 32 operand bytes of a 64-deep block-scaled MFMA: the fragments of two consecutive k-blocks -- ALWAYS the same two register quads per row tile
    // (Tl[i] | Th[i]), so that the pair is one aligned 8-register tuple and no copy is needed (gemm_bf16w_kernel alternates three fragment sets and pays for it)
    auto cat = [](const v4i& a, const v4i& b) { return v8i{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; };
    auto catb = [](const bf16x8& lo, const bf16x8& hi) {
        const v4i a = __builtin_bit_cast(v4i, lo), b = __builtin_bit_cast(v4i, hi);
        return v8i{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };

    // ---- epilogue pieces -----------------------------------------------------------------------------------------------------------------
    // the tile being stored (the PREVIOUS one in the steady state)
    int pm0 = 0, pn0 = 0, ppar = 0;
    bool pvalid = false;                                          // false for the workgroup's first tile: its "previous" tile's stores fall outside an EMPTY buffer resource
    const int rr = lane / LPR, rc = lane % LPR;
    // W(ph): this wave's 32 x 32 block of row tile ph -> image
    auto epi_w = [&](int ph, const f32x16& acc) {
        const float* sc = scb + ppar * SC_FLOATS;
        const float sa = sc[ph * 32 + l31];
        const int ir = l31;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int lc = wave * 32 + 8 * qd + 4 * lh;
            const f32x4 sw4 = *reinterpret_cast<const f32x4*>(sc + 128 + lc);
            const f32x4 add = *reinterpret_cast<const f32x4*>(sc + 256 + lc);
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = acc[4 * qd + c];
                t = t * (sa * sw4[c]) + add[c];
                if constexpr (ACT == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                else if constexpr (ACT == MMDM_EPI_BIAS_SILU) t = silu(t);
                v[c] = t;
            }
            const int cb = lc * EBO;
            char* dst = img + ir * RB + (((cb >> 4) ^ (ir & SWM)) << 4) + (cb & 15);
            if constexpr (OUT == 1) { const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; *reinterpret_cast<bf16x4*>(dst) = o; }
            else if constexpr (OUT == 2) *reinterpret_cast<unsigned*>(dst) = pack_fp8x4(v * p.out_scale);
            else *reinterpret_cast<f32x4*>(dst) = v;
        }
    };
    // R(ph): image -> C, whole rows (this wave: rows 8 wave .. 8 wave + 7 of the phase), in two parts: the image rows are READ before the step's
    // barrier (which then frees the image for the next phase) and STORED behind it -- so that every vector memory operation of the epilogue is
    // issued after the step's counted wait, i.e. strictly behind the step's operand requests, whatever the scheduler does inside a half step
    // (the counted waits rely on that order: see step()).
    f32x4 rv[NR];
    auto epi_r_read = [&]() {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int ir = wave * 8 + k * RPI + rr;
            rv[k] = *reinterpret_cast<const f32x4*>(img + ir * RB + ((rc ^ (ir & SWM)) << 4));
        }
    };
    auto epi_r_store = [&](int ph) {
        const int rows_here = pvalid ? min(p.M - pm0, BM) : 0;
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(p.C) + (size_t)pm0 * p.ldc * EBO, 0, rows_here * p.ldc * EBO, 0x00020000);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int ir = wave * 8 + k * RPI + rr;
            f32x4 v = rv[k];
            if constexpr (EXT) {
                const int rowc = min(pm0 + ph * 32 + ir, p.M - 1);
                const int er = p.epilogue == MMDM_EPI_BIAS_PE ? rowc % p.period : rowc;
                v += *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + pn0 + rc * 4);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (ph * 32 + ir) * p.ldc * EBO + pn0 * EBO + rc * 16, 0, 2);
        }
    };

    float sc1 = 0.f, sc2 = 0.f;                               // a tile's scale / bias operands between their request and the step that parks them
    // per-thread constants of those requests (thread t < 128: a_scale of the tile's row t, or a_const; t >= 128: bias of column t - 128)
    const bool sc_rowsel = tid < 128 && p.a_scale != nullptr, sc_const = tid < 128 && p.a_scale == nullptr;
    const float* const sc_base1 = tid < 128 ? (p.a_scale ? p.a_scale : p.w_scale) : p.bias;
    const int sc_off1 = tid < 128 ? (p.a_scale ? tid : 0) : tid - 128, sc_dst1 = tid < 128 ? tid : tid + 128;
    v4i Tl[TM], Th[TM];                                       // the A fragments in flight: ONE set (k-blocks 0 | 1, then 2 | 3 of the step), refilled behind the MFMA that read it
    // ---- one K step.  POS: position in its group of 8; EPI: the previous tile's epilogue chunk of this position is issued; ZERO: the step's first
    // MFMAs start the accumulators; XP: the stores the PREVIOUS step issued last (its R chunk) -- the counted wait leaves them in flight.
    // On entry (Tl | Th) hold k-blocks 0 | 1 of stage `cur` and b this step's W fragments.  Every MFMA is followed by the two reads that refill ITS A
    // fragment pair (k-blocks 2 | 3 behind the first group, k-blocks 0 | 1 of the next stage behind the second, behind the barrier that publishes
    // it) and by ONE operand request: a vector memory instruction costs its wave 60-180 issue cycles (MI355X_MICROARCH.md), so they go one per MFMA
    // (eight in a row behind one MFMA: QKV 75.2 us; spread: 68.6 us against 70.8 of gemm_bf16w_kernel on the same box).  What is requested:
    //   shipped (DEEP = 0): first group: the W fragments of step kt + 1 into the OTHER register set (bo); second group: stage kt + 2 of the
    //      three-stage ring; the counted wait needs stage kt + 1, requested by step kt - 1's second group.
    //   DEEP = 1: the W fragments a group has just consumed are re-requested for step kt + 2 INTO THE SAME REGISTERS (two per group: two steps to
    //      arrive instead of one), stage kt + 3 of a four-stage ring (two pieces per group).  Built, bit-identical, and measured NO faster (QKV 72.4
    //      vs 68.6 us, LAB_NOTES.md round 6): the loop does not wait for memory -- with all-zero operands (the same instruction stream at a
    //      fraction of the matrix cores' power) either form runs 24 % faster: what bounds the real launch is the clock the power budget allows.
    auto step = [&](auto pos_c, auto epi_c, auto zero_c, auto xp_c, auto sc_c, f32x16 (&acc)[TM], const f32x16 (&accp)[TM], int cur, int nxt, int nfill,
                    int vw_b, int kt_b, __amdgpu_buffer_rsrc_t rs_s, int kt_s, int sc_m0, int sc_n0, int sc_par, bf16x8 (&b)[NKB], bf16x8 (&bo)[NKB]) {
        constexpr int POS = decltype(pos_c)::value;
        constexpr bool EPI = decltype(epi_c)::value != 0, ZERO = decltype(zero_c)::value != 0;
        constexpr int XP = decltype(xp_c)::value;
        constexpr int SC = decltype(sc_c)::value;                 // 1: this step requests the tile's scale / bias operands (two registers until ...); 2: ... this step parks them in LDS
        if constexpr (EPI) {
            if constexpr ((POS & 1) == 0) epi_w(POS >> 1, accp[POS >> 1]);
            else epi_r_read();
        }
        if constexpr (SC == 2) {                                  // (requested two steps ago: the counted waits in between have seen them land)
            float* sc = scb + sc_par * SC_FLOATS;
            sc[sc_dst1] = sc_const ? p.a_const : sc1;                                              // a_scale[row t] | bias[column t - 128]
            sc[128 + (tid & 127)] = sc2;                                                           // w_scale (written twice with the same value)
        }
        const v8i b01 = catb(b[0], b[1]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (ZERO) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b01, cat(Tl[i], Th[i]), z, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            } else
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b01, cat(Tl[i], Th[i]), acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            Tl[i] = rda1(cur, 2, i);
            Th[i] = rda1(cur, 3, i);
        }
        if constexpr (DEEP) { ldbh(vw_b, kt_b, 0, b); stageh(nfill, rs_s, kt_s, 0); }
        else { ldbh(vw_b, kt_b, 0, bo); ldbh(vw_b, kt_b, 1, bo); }
#pragma unroll
        for (int i = 0; i < TM; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        // in flight behind the pieces of stage kt + 1: the previous step's stores and this group's 4 requests (DEEP: + the 8 requests of step kt - 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEEP ? NV : 0) + NV / 2 + XP) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const v8i b23 = catb(b[2], b[3]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b23, cat(Tl[i], Th[i]), acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            Tl[i] = rda1(nxt, 0, i);
            Th[i] = rda1(nxt, 1, i);
        }
        if constexpr (DEEP) { ldbh(vw_b, kt_b, 1, b); stageh(nfill, rs_s, kt_s, 1); }
        else { stageh(nfill, rs_s, kt_s, 0); stageh(nfill, rs_s, kt_s, 1); }
        // the step's EXTRAS, behind its requests: the tile's scale / bias operands, then -- pinned last, so that the next step's wait may count on them
        // being younger than every request of this step -- the R chunk's stores
        if constexpr (SC == 1) {                                  // thread t: a_scale of row t (t < 128) or bias of column t - 128; w_scale of column t & 127
            sc1 = sc_base1[(sc_rowsel ? sc_m0 : sc_n0) + sc_off1];  // (addresses selected, loads unconditional: a branch here would cut the step into blocks,
            sc2 = p.w_scale[sc_n0 + (tid & 127)];                   //  and MFMAs sink across block boundaries -- and barriers -- to their first use)
        }
        if constexpr (EPI && (POS & 1)) epi_r_store(POS >> 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x100, 2, 1); __builtin_amdgcn_sched_group_barrier(0x020, 1, 1); }
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 1);        // (extras that READ: in no fixed order among the requests -- XP never counts them)
        __builtin_amdgcn_sched_group_barrier(0x040, 8, 1);        // the stores: behind every read of this half step
        __builtin_amdgcn_sched_barrier(0);
    };

    bf16x8 bx[NKB], by[NKB];
    f32x16 accA[TM], accB[TM];
    int cur = 0;

    // One tile's K loop (NG groups of 8 steps) into `acc`, with the previous tile's epilogue (from `accp`) under its first group.
    // (cm0, cn0): this tile; (nm0, nn0): the next one (the last two steps request its first stages / fragments; a workgroup's last tile
    // re-requests its own -- valid addresses, never used).
    auto body = [&](f32x16 (&acc)[TM], const f32x16 (&accp)[TM], int cm0, int cn0, int cpar, int nm0, int nn0) {
#ifndef FP8P_EPI
#define FP8P_EPI 1              // 0: experiment builds of tools/ only -- the K loops alone (no epilogue chunk under them; results wrong)
#endif
        constexpr int EPI = FP8P_EPI;
        constexpr int XP0 = NG == 1 ? NS : 0;                     // the previous body's last step issued an R chunk (the prologue drains its requests, so the first tile may wait loosely too)
        const __amdgpu_buffer_rsrc_t rsc = rsA_of(cm0), rsn = rsA_of(nm0);
        const int vwc = voffW_of(cn0), vwn = voffW_of(nn0);
        if constexpr (EPI) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(const_cast<f32x16&>(accp[0])), "+v"(const_cast<f32x16&>(accp[1])), "+v"(const_cast<f32x16&>(accp[2])), "+v"(const_cast<f32x16&>(accp[3])));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const bool lastg = g == NG - 1;
            const int k0 = 8 * g;
            // even / odd steps swap the fragment and W register sets (f0 <-> f2, b <-> bn)
#define FP8P_STEP(POS, E, Z, XP, SC, B0, B1)                                                                                                           \
            {                                                                                                                                        \
                constexpr int DW = DEEP ? 2 : 1, DA = DEEP ? 3 : 2;                  /* steps ahead: W fragments, A stage */                          \
                const int c1 = cur == NSTG - 1 ? 0 : cur + 1, cf = cur == 0 ? NSTG - 1 : cur - 1;       /* next stage; the one being refilled = cur + NSTG - 1 */ \
                const bool nb = lastg && (POS) + DW >= 8, ns = lastg && (POS) + DA >= 8;        /* the NEXT tile's */                                    \
                step(IC<POS>{}, IC<E>{}, IC<Z>{}, IC<XP>{}, IC<SC>{}, acc, accp, cur, c1, cf, nb ? vwn : vwc, nb ? (POS) + DW - 8 : k0 + (POS) + DW,   \
                     ns ? rsn : rsc, ns ? (POS) + DA - 8 : k0 + (POS) + DA, cm0, cn0, cpar, B0, B1);                                                   \
                cur = c1;                                                                                                                            \
            }
            if (g == 0) {
                constexpr int E = EPI;
                constexpr int X = E ? NS : 0;                                   // extras of an odd step of this group
                FP8P_STEP(0, E, 1, XP0, 0, bx, by)
                FP8P_STEP(1, E, 0, 0, 1, by, bx)
                FP8P_STEP(2, E, 0, X, 0, bx, by)
                FP8P_STEP(3, E, 0, 0, 2, by, bx)
                FP8P_STEP(4, E, 0, X, 0, bx, by)
                FP8P_STEP(5, E, 0, 0, 0, by, bx)
                FP8P_STEP(6, E, 0, X, 0, bx, by)
                FP8P_STEP(7, E, 0, 0, 0, by, bx)
            } else {
                FP8P_STEP(0, 0, 0, (EPI ? NS : 0), 0, bx, by)
                FP8P_STEP(1, 0, 0, 0, 0, by, bx)
                FP8P_STEP(2, 0, 0, 0, 0, bx, by)
                FP8P_STEP(3, 0, 0, 0, 0, by, bx)
                FP8P_STEP(4, 0, 0, 0, 0, bx, by)
                FP8P_STEP(5, 0, 0, 0, 0, by, bx)
                FP8P_STEP(6, 0, 0, 0, 0, bx, by)
                FP8P_STEP(7, 0, 0, 0, 0, by, bx)
            }
#undef FP8P_STEP
        }
    };
    // a tile's epilogue on its own (the workgroup's last tile): the four phases with their own barriers
    auto drain = [&](const f32x16 (&accp)[TM]) {
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(const_cast<f32x16&>(accp[0])), "+v"(const_cast<f32x16&>(accp[1])), "+v"(const_cast<f32x16&>(accp[2])), "+v"(const_cast<f32x16&>(accp[3])));
#pragma unroll
        for (int ph = 0; ph < TM; ++ph) {
            epi_w(ph, accp[ph]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            epi_r_read();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            epi_r_store(ph);
        }
    };

    // ---- prologue: the first tile's first two stages and fragments ---------------------------------------------------------------------------
    int cm0, cn0, nm0, nn0;
    tile_mn(0, cm0, cn0);
    {
        const __amdgpu_buffer_rsrc_t rs0 = rsA_of(cm0);
        const int vw0 = voffW_of(cn0);
#pragma unroll
        for (int k = 0; k < NSTG - 1; ++k) { stageh(k, rs0, k, 0); stageh(k, rs0, k, 1); }   // the first stages (K >= 1024: at least 8 steps)
        ldbh(vw0, 0, 0, bx); ldbh(vw0, 0, 1, bx);
        if constexpr (DEEP) { ldbh(vw0, 1, 0, by); ldbh(vw0, 1, 1, by); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (everything: the first steps' counted waits assume a full pipeline behind them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TM; ++i) { Tl[i] = rda1(0, 0, i); Th[i] = rda1(0, 1, i); }
    }
    int j = 0, par = 0;
    tile_mn(1, nm0, nn0);
#ifdef FP8P_TL              // diagnostic builds (tools/mk_variant.sh ... -DFP8P_TL; tools/fp8p_timeline.py): shader-clock stamps per tile, kept in LDS, dumped at the end
    unsigned long long* const tls = reinterpret_cast<unsigned long long*>(scb + 2 * SC_FLOATS);
    const unsigned long long t_r0 = __builtin_amdgcn_s_memrealtime();
    FP8P_STAMP(0);
    // (no branch: a divergent `if` in the loop cuts the step into blocks and the MFMAs sink to their uses -- every lane writes, lanes != 0 into a trash row)
#define FP8P_STAMP(k) do { tls[tid == 0 ? ((k) < 30 ? (k) : 31) : 32 + tid] = __builtin_readcyclecounter(); } while (0)
#else
#define FP8P_STAMP(k) do { } while (0)
#endif
    auto advance = [&]() {                                               // the tile just computed becomes the one to store; false after the workgroup's last tile
        pm0 = cm0; pn0 = cn0; ppar = par; pvalid = true;
        ++j; par ^= 1;
        cm0 = nm0; cn0 = nn0;
        tile_mn(j + 1, nm0, nn0);
        return j < ntl;
    };
#pragma nounroll
    for (;;) {
        body(accA, accB, cm0, cn0, par, nm0, nn0);
        FP8P_STAMP(j + 1);
        if (!advance()) break;
        body(accB, accA, cm0, cn0, par, nm0, nn0);
        FP8P_STAMP(j + 1);
        if (!advance()) {
#pragma unroll
            for (int i = 0; i < TM; ++i) accA[i] = accB[i];             // (once per workgroup: one drain for either parity)
            break;
        }
    }
    drain(accA);
#ifdef FP8P_TL
    if (tid == 0 && p.tl) {
        const unsigned long long t_r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = p.tl + 32 * (size_t)blockIdx.x;
        for (int k = 0; k < 30; ++k) o[k] = k <= ntl ? tls[k] : 0;       // [0] start of the first tile's body, [k] end of tile k's body
        o[30] = __builtin_readcyclecounter();                             // after the drain
        o[31] = ((t_r1 - t_r0) << 32) | (unsigned)ntl;                    // 100 MHz ticks of the whole walk | tiles
    }
#endif
#endif
}

template <int OUT, int ACT, bool EXT, int NG>
int launch_p(const Fp8pArgs& a, hipStream_t st, int grid) {
    mmdm_note_gemm("gemm_fp8p<%d,%d,%d,%d>", OUT, ACT, (int)EXT, NG);
    hipLaunchKernelGGL((gemm_fp8p_kernel<OUT, ACT, EXT, NG>), dim3(grid), dim3(256), SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_fp8p");
}

unsigned long long* g_fp8p_tl = nullptr;     // mmdm_diag_set "fp8p_timeline": 32 u64 per workgroup of the next launches (diagnostic build only)
int g_fp8p_grid = 512;          // mmdm_diag_set "fp8p_grid": workgroups of a launch (a multiple of 8; two per CU by default)

}  // namespace

bool mmdm_fp8p_covers(const Fp8pArgs& a) {
    if (!a.A || !a.W || !a.C || !a.w_scale || !a.bias) return false;
    if ((a.M % 128) || (a.N % 128) || !(a.K == 1024 || a.K == 2048)) return false;
    const bool ext = a.epilogue == MMDM_EPI_BIAS_RESID || a.epilogue == MMDM_EPI_BIAS_PE;
    if (ext || a.out_mode == 0) return false;                     // fp32 output (+ residual): FP8P_EXT below
    if (a.out_mode != 0 && a.epilogue == MMDM_EPI_BIAS_SILU) return false;
    if ((long)(a.M / 128) * (a.N / 128) < 512) return false;      // fewer tiles than workgroups: nothing to overlap
    return true;
}

int mmdm_fp8p_init(void) {
    for (const void* f : {reinterpret_cast<const void*>(&gemm_fp8p_kernel<1, MMDM_EPI_BIAS, false, 1>), reinterpret_cast<const void*>(&gemm_fp8p_kernel<2, MMDM_EPI_BIAS_GELU, false, 1>),
                          reinterpret_cast<const void*>(&gemm_fp8p_kernel<2, MMDM_EPI_BIAS, false, 1>), reinterpret_cast<const void*>(&gemm_fp8p_kernel<1, MMDM_EPI_BIAS_GELU, false, 1>)
                          }) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_fp8p): %s", hipGetErrorString(e));
    }
    return MMDM_OK;
}

int mmdm_fp8p_launch(Fp8pArgs a, hipStream_t st) {
    a.mt = a.M / 128; a.nt = a.N / 128; a.ntiles = a.mt * a.nt;
    a.tl = g_fp8p_tl;
    const int grid = g_fp8p_grid;
    const bool ext = a.epilogue == MMDM_EPI_BIAS_RESID || a.epilogue == MMDM_EPI_BIAS_PE;
    const bool gelu = a.epilogue == MMDM_EPI_BIAS_GELU;
    if (ext || a.out_mode == 0) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "gemm_fp8p: fp32 output is not covered");
    if (a.K != 1024) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "gemm_fp8p: K = %d with a 16-bit / fp8 output", a.K);
    if (a.out_mode == 1) return gelu ? launch_p<1, MMDM_EPI_BIAS_GELU, false, 1>(a, st, grid) : launch_p<1, MMDM_EPI_BIAS, false, 1>(a, st, grid);
    return gelu ? launch_p<2, MMDM_EPI_BIAS_GELU, false, 1>(a, st, grid) : launch_p<2, MMDM_EPI_BIAS, false, 1>(a, st, grid);
}

bool mmdm_diag_gemm_fp8p(const char* key, long long v) {
    if (!strcmp(key, "fp8p_timeline")) { g_fp8p_tl = reinterpret_cast<unsigned long long*>((uintptr_t)v); return true; }
    if (!strcmp(key, "fp8p_grid")) { if (v >= 8 && v <= 4096 && !(v & 7)) g_fp8p_grid = (int)v; return true; }
    return false;
}
