// Exact-fp32 linear layer for gfx950: C = A W^T + b with fused epilogues, on v_mfma_f32_32x32x2_f32.
//
// Replaces torch.nn.functional.linear at every call site of the denoising path (SURVEY.md 2.3 K1,K2,K4,K6,K7,K8;
// reference: src/models/utils/layers.py:33-34,74-75,99-106,109-116, src/models/in2in.py:389-390,426-431).
//
// Tiling (64-wide wavefronts): 128x128 output tile per 256-thread workgroup, 4 waves as 2(M) x 2(N), each wave
// 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  K is walked in steps of 32 through a
// double-buffered LDS image; rows are padded to 36 floats so that every ds_read_b128 lane group touches 16
// distinct 16-byte bank slots (36*i mod 64 is a distinct multiple of 4 for 16 distinct i mod 16).
// K order inside a step is permuted identically for both operands (lane half h supplies k = 8g + 4h + s for MFMA
// s = 0..3 of group g), which lets each lane fetch its four operands for four MFMAs with one 16-byte LDS read.
// Global->LDS staging is register-staged (global_load_dwordx4 -> ds_write_b128): the padded image rules out LDS-DMA.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles that share A rows.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A; const float* W; const float* bias; float* C; const float* extra;
    int lda, ldw, ldc, ld_extra;
    int M, N, K, Kw, epilogue, period;   // Kw >= K: readable columns of W (zero beyond K)
    int mt, nt;
    int ablate;                          // timing experiments only (tools/gemm_bench.py); 0 in production
    unsigned long long* stamps;          // diagnostic builds of a launch only (tools/gemm_timeline.py): per-workgroup {start, loop start, loop end, placement, kernel end, kernel entry, residual landed, stores issued}, then {s_memtime at loop start, loop end} per workgroup
                                         // in 100 MHz s_memrealtime ticks; nullptr in production (one never-taken scalar branch per workgroup)
};

template <int VEC, bool FULL>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ base, int ld, int row, int nrows, int k, int K) {
    if (FULL) return *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + k);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) {
        const float* p = base + (size_t)row * ld + k;
        if (VEC == 4) {
            if (k < K) v = *reinterpret_cast<const f32x4*>(p);
        } else {
            if (k + 0 < K) v.x = p[0];
            if (k + 1 < K) v.y = p[1];
            if (k + 2 < K) v.z = p[2];
            if (k + 3 < K) v.w = p[3];
        }
    }
    return v;
}


// Tile configuration: the workgroup is WGM x WGN waves (encoded as TM = 10*WGM + tiles, TN likewise: TM=22 -> 2 waves x 2 tiles),
// each wave owns (TM%10) x (TN%10) MFMA tiles of 32x32; K step BK.
template <int TM_, int TN_, int BK_>
struct Cfg {
    static constexpr int WGM = TM_ >= 10 ? TM_ / 10 : 2, WGN = TN_ >= 10 ? TN_ / 10 : 2;
    static constexpr int TM = TM_ % 10, TN = TN_ % 10;
    static constexpr int THREADS = 64 * WGM * WGN;
    static constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = BK_, LDP = BK_ + 4;
    static constexpr int A_FLOATS = BM * LDP, B_FLOATS = BN * LDP;
    static constexpr int SMEM_BYTES = 2 * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int F4_PER_ROW = BK_ / 4, ROWS_PER_PASS = THREADS / F4_PER_ROW;
    static constexpr int PA = BM / ROWS_PER_PASS, PB = BN / ROWS_PER_PASS;
    static constexpr int G = BK_ / 8;                       // groups of 8 k (4 MFMA k-steps) per K step
};

template <int TM_, int TN_, int BK_, int AVEC, int WVEC, bool FULL>
__global__ __launch_bounds__((Cfg<TM_, TN_, BK_>::THREADS)) void gemm_f32_kernel(GemmArgs p) {
    using C_ = Cfg<TM_, TN_, BK_>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, LDP = C_::LDP, TM = C_::TM, TN = C_::TN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM*LDP]
    float* Bs = smem + 2 * C_::A_FLOATS;       // [2][BN*LDP]

    // XCD-aware remap (bijective for any grid size): blocks b and b+8 share an XCD; give each XCD a contiguous tile range.
    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (swz / p.nt) * BM;
    const int n0 = (swz % p.nt) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C_::WGN, wn = wave % C_::WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    // staging map: thread -> rows (tid / F4_PER_ROW) + ROWS_PER_PASS*j, float4 column tid % F4_PER_ROW
    const int sr = tid / C_::F4_PER_ROW, sc = (tid % C_::F4_PER_ROW) * 4;

    // Accumulators start as bias (+ residual / + PE row): the epilogue's extra reads are issued here, where their latency
    // overlaps the first operand loads and the co-resident workgroups' MFMAs, instead of after the last MFMA.
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + j * 32 + l31;
            const bool cok = FULL || col < p.N;
            const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = bv;
                if (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) {
                    const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (cok && (FULL || row < p.M)) {
                        const int er = p.epilogue == MMDM_EPI_BIAS_PE ? row % p.period : row;
                        v += p.extra[(size_t)er * p.ld_extra + col];
                    }
                }
                acc[i][j][e] = v;
            }
        }

    const int nkt = (p.K + BK - 1) / BK;
    f32x4 ra[C_::PA], rb[C_::PB];
#pragma unroll
    for (int j = 0; j < C_::PA; ++j) ra[j] = load4<AVEC, FULL>(p.A, p.lda, m0 + sr + C_::ROWS_PER_PASS * j, p.M, sc, p.K);
#pragma unroll
    for (int j = 0; j < C_::PB; ++j) rb[j] = load4<WVEC, FULL>(p.W, p.ldw, n0 + sr + C_::ROWS_PER_PASS * j, p.N, sc, p.Kw);
#pragma unroll
    for (int j = 0; j < C_::PA; ++j) *reinterpret_cast<f32x4*>(&As[(sr + C_::ROWS_PER_PASS * j) * LDP + sc]) = ra[j];
#pragma unroll
    for (int j = 0; j < C_::PB; ++j) *reinterpret_cast<f32x4*>(&Bs[(sr + C_::ROWS_PER_PASS * j) * LDP + sc]) = rb[j];
    __syncthreads();

    const int a_off = (wm * (32 * TM) + l31) * LDP + 4 * lh;
    const int b_off = (wn * (32 * TN) + l31) * LDP + 4 * lh;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nkt;
        if (more && p.ablate < 1) {
            const int k0 = (kt + 1) * BK + sc;
#pragma unroll
            for (int j = 0; j < C_::PA; ++j) ra[j] = load4<AVEC, FULL>(p.A, p.lda, m0 + sr + C_::ROWS_PER_PASS * j, p.M, k0, p.K);
#pragma unroll
            for (int j = 0; j < C_::PB; ++j) rb[j] = load4<WVEC, FULL>(p.W, p.ldw, n0 + sr + C_::ROWS_PER_PASS * j, p.N, k0, p.Kw);
        }
        const float* Ac = As + (p.ablate >= 1 ? 0 : cur) * C_::A_FLOATS + a_off;
        const float* Bc = Bs + (p.ablate >= 1 ? 0 : cur) * C_::B_FLOATS + b_off;
        // fragments of group g+1 are fetched before the MFMAs of group g (register double buffering)
        f32x4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDP);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDP);
#pragma unroll
        for (int g = 0; g < C_::G; ++g) {
            const int cb = g & 1, nb = cb ^ 1;
            if (g + 1 < C_::G) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[nb][i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDP + (g + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[nb][j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDP + (g + 1) * 8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i][s], bf[cb][j][s], acc[i][j], 0, 0, 0);
        }
        if (more && p.ablate < 1) {
            float* An = As + (cur ^ 1) * C_::A_FLOATS;
            float* Bn = Bs + (cur ^ 1) * C_::B_FLOATS;
#pragma unroll
            for (int j = 0; j < C_::PA; ++j) *reinterpret_cast<f32x4*>(&An[(sr + C_::ROWS_PER_PASS * j) * LDP + sc]) = ra[j];
#pragma unroll
            for (int j = 0; j < C_::PB; ++j) *reinterpret_cast<f32x4*>(&Bn[(sr + C_::ROWS_PER_PASS * j) * LDP + sc]) = rb[j];
        }
        if (p.ablate < 2) __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + j * 32 + l31;
            if (!FULL && col >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (!FULL && row >= p.M) continue;
                float v = acc[i][j][e];
                if (p.epilogue == MMDM_EPI_BIAS_GELU) v = gelu_erf(v);
                else if (p.epilogue == MMDM_EPI_BIAS_SILU) v = silu(v);
                else if (p.epilogue == MMDM_EPI_BIAS_QUICKGELU) v = quick_gelu(v);
                else if (p.epilogue == MMDM_EPI_BIAS_SIGMOID) v = sigmoidf(v);
                p.C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant (the production path when K % 16 == 0 and operands are 16-byte aligned).
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write): each wave-instruction
// lands 1 KiB = 16 rows x 64 B contiguously, so the LDS image is unpadded [row][16 floats]; bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle applied on the per-lane SOURCE address and again on the
// read (16-byte chunk c of row r is stored at chunk c ^ ((r >> 2) & 3)): a 16-lane read group then covers 16 distinct
// 16-byte bank slots.  Rows past M / N are clamped on load (their results are never stored).
// ---------------------------------------------------------------------------------------------------------
template <int TM_, int TN_, int BK_ = 16, int NBUF_ = 2>
struct GCfg {
    static constexpr int WGM = TM_ / 10, WGN = TN_ / 10, TM = TM_ % 10, TN = TN_ % 10;
    static constexpr int NWAVES = WGM * WGN, THREADS = 64 * NWAVES;
    static constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = BK_;
    static constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK;
    static constexpr int NBUF = NBUF_;
    static constexpr int SMEM_BYTES = NBUF * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int CPR = BK / 4;                               // 16-byte chunks per row (4 or 8)
    static constexpr int RPP = 64 / CPR;                             // rows per 1-KiB piece (16 or 8)
    static constexpr int NA = BM / RPP, NB = BN / RPP;               // pieces per operand tile
    static constexpr int NI = (NA + NB) / NWAVES;                    // pieces per wave (exact)
    static_assert((NA + NB) % NWAVES == 0, "pieces must divide evenly over the waves (counted vmcnt)");
    static constexpr int G = BK / 8;                                 // groups of 8 k per K step
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// swizzle: 16-byte chunk c of tile row r is stored at chunk position c ^ swz(r)
template <int CPR> __device__ __forceinline__ int swz_of(int r) { return CPR == 4 ? ((r >> 2) & 3) : ((r >> 1) & 7); }

// VEPI: the MFMA operands are swapped (D^T = W A^T), which puts the output ROW on the lane and four consecutive output
// COLUMNS in consecutive accumulator registers -- bias / residual / PE loads and the result stores are then 16-byte
// accesses (4x fewer memory instructions than the one-float-per-lane map).  Needs N % 4 == 0 and 16-byte aligned C / extra rows.
// MINW_: minimum waves per SIMD the register allocation must allow (second __launch_bounds__ argument; 1 = no constraint).  The 64
// accumulator registers of a 64x64 wave tile plus operands, addresses and the accumulator-initialisation transients come to ~144
// registers when unconstrained (3 waves per SIMD = THREE 4-wave workgroups per CU); MINW_ = 5 makes the compiler fit 96.
//
// PIPE_ = 1: software-pipelined K loop for NBUF_ >= 3 stages (the production loop).  The 2-stage loop above it makes every K step pay
// its own latencies in sequence -- vmcnt(0) on a tile issued only one step (2048 MFMA cycles) earlier, the barrier, the LDS-DMA issue
// of the next tile, then the fragment reads -- before its first MFMA can issue: a workgroup alone on a CU reaches ~70 % of the MFMA rate
// and the kernel depends on co-resident workgroups to fill the holes (three fit: 144 registers; tools/gemm_timeline.py).  Here
//   * tile kt+NBUF-1 is issued during step kt (NBUF-2 whole steps of latency budget), its DMA instructions interleaved one by one
//     behind MFMAs (an LDS-DMA issue costs ~60-100 cycles of the wave's issue stream: it hides in the 64-cycle shadow of an MFMA);
//   * the fragments of k-group 1 are read at the top of the step and those of the NEXT tile's k-group 0 in the middle of it, so every
//     ds_read has 16 MFMAs (1024 cycles) to land and no MFMA ever waits on LDS;
//   * the one wait + barrier per step sits between the two MFMA groups, where the matrix pipe still holds queued work.
// Hazards: a buffer is restaged only after a barrier that every wave reached with lgkmcnt(0) behind its last read of that buffer; a
// staged tile is read only after each wave's counted vmcnt has retired its own pieces of it AND the barrier (guide: "read a staged
// buffer after the wait that retires it and a barrier the reader has passed").
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One output tile (`swz` = tile index after the XCD remap) from start to finish.  Device pass only: the buffer-resource type of the
// LDS-DMA builtins does not exist in the host pass, which only needs the kernel's symbol.
#if defined(__HIP_DEVICE_COMPILE__)
// DIAG_: the timing ablations (GemmArgs::ablate) and in-kernel stamps (GemmArgs::stamps) exist in a second instantiation only, launched
// when a tool has set one of them (tools/gemm_bench.py ABL=, tools/gemm_timeline.py, bench.py's loop clock); the production instantiation
// sees compile-time zeros -- no diagnostic branch, load or register in the shipped kernels.
template <int TM_, int TN_, int BK_, int NBUF_, bool VEPI, int PIPE_, bool DIAG_>
__device__ __forceinline__ void gemm_tile(const GemmArgs& pp, float* smem, int swz, int bid) {
    using C_ = GCfg<TM_, TN_, BK_, NBUF_>;
    const GemmArgs& p = pp;
    unsigned long long* const p_stamps = DIAG_ ? pp.stamps : nullptr;
    const int p_ablate = DIAG_ ? pp.ablate : 0;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TM = C_::TM, TN = C_::TN, CPR = C_::CPR, RPP = C_::RPP, NBUF = C_::NBUF;
    float* As = smem;                                   // [NBUF][BM*BK]
    float* Bs = smem + NBUF * C_::A_FLOATS;             // [NBUF][BN*BK]

    const int m0 = (swz / p.nt) * BM;
    const int n0 = (swz % p.nt) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C_::WGN, wn = wave % C_::WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    if (p_stamps && tid == 0) {
        p_stamps[8 * (size_t)bid + 0] = __builtin_amdgcn_s_memrealtime();
        p_stamps[8 * (size_t)bid + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                        ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) << 32);   // HW_ID, XCC_ID
    }

    // Staging pieces of this wave: piece id pq = wave + NWAVES*u; pq < NA -> A rows RPP*pq.., else W rows RPP*(pq-NA)..  The pieces are
    // requested with `buffer_load_dwordx4 ... offen lds`: one buffer resource per operand whose base is the tile's first row, a per-lane
    // 32-bit byte offset that is fixed for the whole tile (row * ld + swizzled chunk) and ONE scalar offset that walks K.  Measured
    // (tools/loop_bench3.hip): with per-lane 64-bit pointers (`global_load_lds`, two VALU adds per piece and step, 64-bit address
    // operands) the operand stream costs 12-16 % of the MFMA rate of this loop; in the buffer form 0-2 %.
    constexpr int UA = C_::NA / C_::NWAVES;             // pieces u < UA are A pieces for every wave, the others W pieces
    static_assert(C_::NA % C_::NWAVES == 0, "A pieces must divide evenly over the waves (compile-time operand of a piece)");
    int voff[C_::NI];                                   // byte offset of this lane's 16 bytes inside the operand's row block, at k = 0
    int dst[C_::NI];                                    // LDS float offset of the piece inside buffer 0 (wave-uniform)
#pragma unroll
    for (int u = 0; u < C_::NI; ++u) {
        const int pq = wave + C_::NWAVES * u;
        const int prow = lane / CPR, pc = lane % CPR;
        const bool isa = u < UA;
        const int trow = (isa ? RPP * pq : RPP * (pq - C_::NA)) + prow;         // row inside the operand tile
        const int gch = pc ^ swz_of<CPR>(trow);                                  // source chunk stored at LDS chunk pc of this row
        if (isa) {
            const int rel = min(trow, p.M - 1 - m0);                            // rows past M re-read the last row (never stored)
            voff[u] = (rel * p.lda + 4 * gch) * 4;
            dst[u] = RPP * pq * BK;
        } else {
            const int rel = min(trow, p.N - 1 - n0);
            voff[u] = (rel * p.ldw + 4 * gch) * 4;
            dst[u] = NBUF * C_::A_FLOATS + RPP * (pq - C_::NA) * BK;
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + (size_t)m0 * p.lda), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W + (size_t)n0 * p.ldw), 0, 0xffffffff, 0x00020000);
    int koff = 0;                                       // byte offset of the K tile being requested (scalar)
    auto stage_part = [&](int buf, auto u0c, auto u1c) {
        constexpr int u0 = decltype(u0c)::value, u1 = decltype(u1c)::value;
#pragma unroll
        for (int u = u0; u < u1; ++u) {
            const int boff = u < UA ? buf * C_::A_FLOATS : buf * C_::B_FLOATS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(u < UA ? rsA : rsW, (lptr_t)(smem + dst[u] + boff), 16, voff[u], koff, 0, 0);
        }
        if constexpr (u1 == C_::NI) koff += BK * 4;
    };
    auto stage = [&](int buf) { stage_part(buf, std::integral_constant<int, 0>{}, std::integral_constant<int, C_::NI>{}); };

    const int nkt = p.K / BK;
    // Pipelined kernel, 16-byte epilogue form (residual / PE GEMMs): the accumulators start as the bias only and the residual tile is
    // fetched while the LAST operand tiles are consumed and added after the loop, y = (b + sum_k a_k w_k) + r -- the order of the
    // reference's `x + linear(...)` -- so a workgroup's first MFMA waits for one 16 KB operand tile instead of 64 KB of residual plus
    // NBUF-1 tiles (13-19 us of every workgroup's life, and of every kernel's start, in the round-1 kernel).  Every pipelined
    // instantiation uses this order, so a row's result does not depend on the tile shape that produced it.
    constexpr bool LATE_R = PIPE_ != 0 && VEPI;
    // accumulators start as bias (+ residual / + PE row unless LATE_R)
    f32x16 acc[TM][TN];
    if (VEPI) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * (32 * TM) + i * 32 + l31;
            const bool rok = row < p.M;
            const int er = p.epilogue == MMDM_EPI_BIAS_PE ? row % p.period : row;
            const bool ext = !LATE_R && (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) && rok;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int col = n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (col < p.N) {
                        if (p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + col);
                        if (ext) v += *reinterpret_cast<const f32x4*>(p.extra + (size_t)er * p.ld_extra + col);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][4 * qd + c] = v[c];
                }
        }
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + j * 32 + l31;
            const bool cok = col < p.N;
            const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = bv;
                if (p.epilogue == MMDM_EPI_BIAS_RESID || p.epilogue == MMDM_EPI_BIAS_PE) {
                    const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (cok && row < p.M) {
                        const int er = p.epilogue == MMDM_EPI_BIAS_PE ? row % p.period : row;
                        v += p.extra[(size_t)er * p.ld_extra + col];
                    }
                }
                acc[i][j][e] = v;
            }
        }
    }

    if constexpr (PIPE_ != 0) {
        // The first NBUF-1 operand tiles are requested behind the accumulator-initialisation loads; vmcnt retires in order, so
        // "at most the NBUF-2 newest tiles outstanding" means the initialisation values and tile 0 have landed: the loop starts on
        // tile 0 while the others are still on their way (the host side guarantees nkt >= NBUF).
#pragma unroll
        for (int t = 0; t < NBUF - 1; ++t) stage(t);
        wait_vm<(NBUF - 2) * C_::NI>();
    } else {
        // the accumulator-init loads above must not be counted by the pipeline's vmcnt arithmetic
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (p_stamps && tid == 0) {
        p_stamps[8 * (size_t)bid + 1] = __builtin_amdgcn_s_memrealtime();
        p_stamps[8 * (size_t)gridDim.x + 2 * bid] = __builtin_readcyclecounter();       // s_memtime (shader clocks) beside the 100 MHz stamp: the clock inside the loop
    }
    // fragment read offsets (floats): row*BK + 4*((2g + lh) ^ sw)
    const int sw = swz_of<CPR>(l31);
    const int a_row = (wm * (32 * TM) + l31) * BK;
    const int b_row = (wn * (32 * TN) + l31) * BK;

    f32x4 rv[LATE_R ? TM : 1][LATE_R ? TN : 1][4];               // the late residual tile (LATE_R only)
    if constexpr (PIPE_ != 0) {
        static_assert((C_::G == 2 || C_::G == 4) && NBUF >= 3 && NBUF <= 6, "pipelined loop: K step 16 or 32 (two / four k-groups), 3 to 6 stages");
        static_assert((C_::NI + 1) / 2 <= 4 * TM * TN - 1, "LDS-DMA pieces of a tile must fit behind the MFMAs of two k-groups");
        constexpr int NI = C_::NI;
        f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
        auto rd = [&](int buf, int g, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
            const int cg = 4 * ((2 * g + lh) ^ sw);
            const float* Ac = As + buf * C_::A_FLOATS + a_row + cg;
            const float* Bc = Bs + buf * C_::B_FLOATS + b_row + cg;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * BK);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * BK);
        };
        auto mm_s = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN], auto s0c, auto s1c) {
#pragma unroll
            for (int s = decltype(s0c)::value; s < decltype(s1c)::value; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = VEPI ? __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        };
        auto mm = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) { mm_s(af, bf, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}); };
        // `left` = tiles that may stay in flight behind the one being waited for (+ the NR residual loads issued at the start of the drain)
        constexpr int NR = LATE_R ? TM * TN * 4 : 0;
        auto wait_left = [&](int left) {
            if (left >= 4) wait_vm<4 * NI + NR>();
            else if (left == 3) wait_vm<3 * NI + NR>();
            else if (left == 2) wait_vm<2 * NI + NR>();
            else if (left == 1) wait_vm<NI + NR>();
            else wait_vm<NR>();
        };
        __builtin_amdgcn_s_barrier();                        // tile 0 has landed for every wave
        rd(0, 0, a0, b0);
        int cur = 0, nxt = 1, stg = NBUF - 1;
        const int n_main = nkt - (NBUF - 1);                 // steps that still issue a new tile
        constexpr int G = C_::G, NMM = 4 * TM * TN;          // k-groups per step (even), MFMAs per group
        // One group: its MFMAs from register set `cs`; behind the FIRST of them the fragment reads of the following group into the other
        // set (the wave's wait for `cs` comes before those reads are issued, so it never waits on them, and they have NMM-1 MFMAs to
        // land); in group 0 of a staging step one LDS-DMA piece behind each of the next MFMAs (at most NMM-2 per group, the rest spill
        // into group 1).  The step's single wait + barrier precedes the LAST group, whose reads are the next tile's group 0.
        auto group = [&](auto gi, auto do_stage, int left, bool more) {
            constexpr int g = decltype(gi)::value;
            constexpr bool STG = decltype(do_stage)::value;
            // the pieces of the tile being requested are split over k-groups 0 and 1 (tools/loop_bench3.hip: 2 + 2 beats 4 + 0)
            constexpr int D0 = (NI + 1) / 2, D1 = NI - D0;
            constexpr int dma_here = (!STG || PIPE_ == 2) ? 0 : (g == 0 ? D0 : (g == 1 ? D1 : 0));
            constexpr int before_wait = G == 2 ? D0 : NI;      // pieces of the newest tile already issued when the step's wait executes
            if constexpr (g == G - 1) {
                if (more) {
                    if constexpr (STG) wait_vm<(NBUF - 3) * NI + before_wait>();      // tile kt+1 landed; tiles kt+2 .. may be in flight
                    else wait_left(left);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (PIPE_ != 3) __builtin_amdgcn_s_barrier();      // PIPE_ 2 / 3: timing ablations (tools/gemm_bench.py), wrong results
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (g % 2 == 0) {
                if constexpr (g < G - 1) rd(cur, g + 1, a1, b1);
                else if (more) rd(nxt, 0, a1, b1);
                if constexpr (STG && g == 0 && PIPE_ != 2) stage_part(stg, std::integral_constant<int, 0>{}, std::integral_constant<int, D0>{});
                mm(a0, b0);
            } else {
                if constexpr (g < G - 1) rd(cur, g + 1, a0, b0);
                else if (more) rd(nxt, 0, a0, b0);
                if constexpr (STG && g == 1 && PIPE_ != 2) stage_part(stg, std::integral_constant<int, D0>{}, std::integral_constant<int, NI>{});
                mm(a1, b1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, g);
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, g);
#pragma unroll
            for (int u = 0; u < dma_here; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, g);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, g);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NMM - 1 - dma_here, g);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto step = [&](auto do_stage, int left, bool more) {
            group(std::integral_constant<int, 0>{}, do_stage, left, more);
            group(std::integral_constant<int, 1>{}, do_stage, left, more);
            if constexpr (G >= 4) {
                group(std::integral_constant<int, 2>{}, do_stage, left, more);
                group(std::integral_constant<int, 3>{}, do_stage, left, more);
            }
            cur = nxt; nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
        };
        for (int kt = 0; kt < n_main; ++kt) {
            step(std::true_type{}, 0, true);
            stg = stg + 1 == NBUF ? 0 : stg + 1;
        }
        if constexpr (LATE_R) {                                          // residual / PE tile: NR 16-byte loads per lane, in flight through the drain
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + wm * (32 * TM) + i * 32 + l31;
                const int er = p.epilogue == MMDM_EPI_BIAS_PE ? row % p.period : row;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int col = n0 + wn * (32 * TN) + j * 32 + 8 * qd + 4 * lh;
                        // rows / columns past the edge read a valid address (row 0 / column 0 of the tile's clamp) and are never stored
                        const int rr = row < p.M ? er : 0, cc = col < p.N ? col : 0;
                        rv[i][j][qd] = *reinterpret_cast<const f32x4*>(p.extra + (size_t)rr * p.ld_extra + cc);
                    }
            }
        }
        for (int kt = n_main < 0 ? 0 : n_main; kt < nkt; ++kt)           // drain: no new tile
            step(std::false_type{}, nkt - kt - 2, kt + 1 < nkt);
    } else {
    stage(0);
    if (NBUF == 3 && nkt > 1) stage(1);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = NBUF == 3 ? kt % 3 : (kt & 1);
        if (NBUF == 3) {
            // tile kt has landed once at most the NI loads of tile kt+1 are still in flight
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C_::NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // raw barrier: __syncthreads() would drain vmcnt(0) while LDS-DMA is in flight and serialise the pipeline
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everyone done reading buffer (kt+2)%3 (tile kt-1)
            if (kt + 2 < nkt && (p_ablate & 7) < 1) stage((kt + 2) % 3);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nkt && (p_ablate & 7) < 1) stage(cur ^ 1);
        }
        const float* Ac = As + ((p_ablate & 7) >= 1 ? 0 : cur) * C_::A_FLOATS + a_row;
        const float* Bc = Bs + ((p_ablate & 7) >= 1 ? 0 : cur) * C_::B_FLOATS + b_row;
#pragma unroll
        for (int g = 0; g < C_::G; ++g) {
            const int cg = 4 * ((2 * g + lh) ^ sw);
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * BK + cg);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * BK + cg);
            if (p_ablate & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = VEPI ? __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            if (p_ablate & 8) __builtin_amdgcn_s_setprio(0);
        }
    }
    }
    if ((p_ablate & 7) >= 3) return;
    // MFMA -> VALU hazard across the loop-exit branch: see MFMA_SETTLE in attn_f32.hip
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[i][j]));
    if (p_stamps && tid == 0) {
        p_stamps[8 * (size_t)bid + 2] = __builtin_amdgcn_s_memrealtime();
        p_stamps[8 * (size_t)gridDim.x + 2 * bid + 1] = __builtin_readcyclecounter();
    }
    if (p_stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) p_stamps[8 * (size_t)bid + 6] = __builtin_amdgcn_s_memrealtime();
    }

    // The activation is chosen ONCE, outside the element loops: with the runtime `p.epilogue` tests inside them every element carried all
    // four activation bodies (11 000 instructions, ~90 KB of code after the loop) and the wave hopped through it branch by branch --
    // 27-47 us per tile beside a co-resident workgroup in its K loop, 8 us when every workgroup ran it at once (tools/gemm_timeline.py).
    // Stores go through a buffer resource over this tile's rows of C: rows past M fall outside `num_records` and are dropped by the
    // hardware (the range check sees the vector offset + immediate, so the row step is added there: one VALU add per store), the
    // column step is an immediate -- no 64-bit address arithmetic and no per-store
    // branch (each `if (row < M)` around a store was an exec-mask branch that the wave had to resolve before its next instruction).
    // Columns past N (only in the last column tile of an N that is not a multiple of the tile) are masked per 32-column block.
    const int rows_here = min(p.M - m0, BM);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)m0 * p.ldc, 0, rows_here * p.ldc * 4, 0x00020000);
    const int ldc4 = p.ldc * 4;
    auto finish = [&](auto act_c, auto fulln_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool FULLN = decltype(fulln_c)::value;
        auto act = [](float t) {
            if constexpr (ACT == MMDM_EPI_BIAS_GELU) return gelu_erf(t);
            else if constexpr (ACT == MMDM_EPI_BIAS_SILU) return silu(t);
            else if constexpr (ACT == MMDM_EPI_BIAS_QUICKGELU) return quick_gelu(t);
            else if constexpr (ACT == MMDM_EPI_BIAS_SIGMOID) return sigmoidf(t);
            else return t;
        };
        if constexpr (VEPI) {
            // D^T map: lane&31 = output row inside the 32-row tile, register 4*qd + c = output column 8*qd + 4*(lane>>5) + c
            const int col0 = n0 + wn * (32 * TN) + 4 * lh;
            const int voff = (wm * (32 * TM) + l31) * ldc4 + col0 * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        f32x4 v;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float t = acc[i][j][4 * qd + c];
                            if constexpr (LATE_R) t += rv[i][j][qd][c];
                            v[c] = act(t);
                        }
                        // the row step is part of the VECTOR offset: only that (plus the immediate) takes part in the range check
                        if (FULLN || col0 + j * 32 + 8 * qd < p.N)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, voff + i * 32 * ldc4 + (j * 32 + 8 * qd) * 4, 0, 0);
                    }
                }
        } else {
            const int col0 = n0 + wn * (32 * TN) + l31;
            const int voff = (wm * (32 * TM) + 4 * lh) * ldc4 + col0 * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!FULLN && col0 + j * 32 >= p.N) continue;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float t = acc[i][j][e];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, act(t)), rsC,
                                                              voff + (i * 32 + (e & 3) + 8 * (e >> 2)) * ldc4 + j * 128, 0, 0);
                    }
                }
        }
    };
    auto finish_n = [&](auto act_c) {
        if (n0 + BN <= p.N) finish(act_c, std::true_type{});
        else finish(act_c, std::false_type{});
    };
    if constexpr (VEPI) {        // the 16-byte form is only launched for the residual / PE epilogues (launch_glds): identity activation
        finish_n(std::integral_constant<int, MMDM_EPI_BIAS>{});
        return;
    }
    switch (p.epilogue) {
        case MMDM_EPI_BIAS_GELU: finish_n(std::integral_constant<int, MMDM_EPI_BIAS_GELU>{}); break;
        case MMDM_EPI_BIAS_SILU: finish_n(std::integral_constant<int, MMDM_EPI_BIAS_SILU>{}); break;
        case MMDM_EPI_BIAS_QUICKGELU: finish_n(std::integral_constant<int, MMDM_EPI_BIAS_QUICKGELU>{}); break;
        case MMDM_EPI_BIAS_SIGMOID: finish_n(std::integral_constant<int, MMDM_EPI_BIAS_SIGMOID>{}); break;
        default: finish_n(std::integral_constant<int, MMDM_EPI_BIAS>{}); break;
    }
}
#endif

template <int TM_, int TN_, int BK_, int NBUF_, bool VEPI, int MINW_ = 1, int PIPE_ = 0, bool DIAG_ = false>
__global__ __launch_bounds__((GCfg<TM_, TN_, BK_, NBUF_>::THREADS), MINW_) void gemm_glds_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    unsigned long long* const stamps = DIAG_ ? p.stamps : nullptr;
    const unsigned long long t_entry = stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    if (stamps && threadIdx.x == 0) stamps[8 * (size_t)bid + 5] = t_entry;
    gemm_tile<TM_, TN_, BK_, NBUF_, VEPI, PIPE_, DIAG_>(p, smem, swz, bid);
    if (stamps) {
        if (threadIdx.x == 0) stamps[8 * (size_t)bid + 7] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) stamps[8 * (size_t)bid + 4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Small launches: the same accumulation chains on 16 x 16 blocks (v_mfma_f32_16x16x4_f32).
//
// An fp32 MFMA accumulator is ONE k-ordered chain of fused multiply-adds (tools/mfma_chain_bits.hip: 32x32x2, 16x16x4 and a scalar fmaf
// chain over the same k order agree in every bit), and a chain of 32 x 32 blocks lasts K / 2 x 64 cycles however few of them a launch has:
// a GEMM of M = 240 rows (configs[0]) against N = 1024 is 64 workgroups of 64 x 64 on 256 CUs, 15.6 us of chain at K = 1024 with three
// quarters of the chip idle.  A 16 x 16 block is a quarter of the chain (32 cycles per 4 k instead of 64 per 2 k) on a quarter of the
// outputs: four times the workgroups, a quarter of the latency.  This kernel walks k in the production order (group g of 8 k: 8g, 8g+4,
// 8g+1, 8g+5 | 8g+2, 8g+6, 8g+3, 8g+7 -- what lane half h = k-slot h of the 32x32x2 operands gives), starts the accumulators as the bias
// and adds the residual / PE row after the loop like the pipelined 16-byte epilogue form: every output element is BIT-IDENTICAL to the
// production kernels' (tests/test_gpu_kernels.py).  Measured (tools/gemm_s16_bench.py, <1 block per wave, 4 stages> vs the 64 x 64 launch):
// 240 x 1024 x 1024: 19.7 -> 13.7 us, x 2048: 36.4 -> 24.0; 6 x 1024 x 1024: 19.9 -> 12.8.  It does NOT carry the reference's B = 1 call
// (M = 1196: 1216 chains of 32 x 32 = two rounds on 1024 matrix pipes, the case DESIGN 10.3 prices): 38.3 -> 35.0 us at N = K = 1024 but
// 80 -> 94 at N = 3072, 50 -> 61 at N = 2048, the step 5.75 -> 6.6 ms -- half the operand reuse per LDS-DMA byte and per fragment read,
// a barrier per 8 MFMAs of a wave.  The dispatch therefore takes it only where the 64 x 64 grid leaves half of the CUs idle.
//
// Workgroup: 32 x (32 TN16) outputs, four waves as 2 x 2, each 16 rows x TN16 blocks of 16 columns; operands swapped (D^T = W A^T) so a lane
// owns an output row and four consecutive columns.  K step 32 through an NBUF-stage LDS-DMA ring (one barrier per step; the request of
// tile kt + NBUF - 1 follows the barrier of step kt; surplus requests of the last steps re-read the last tile into a free stage, so the
// counted vmcnt stays a compile-time number).  128-byte LDS rows, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7).
// ---------------------------------------------------------------------------------------------------------
template <int TN16_, int NBUF_>
struct S16Cfg {
    static constexpr int TN16 = TN16_, NBUF = NBUF_;
    static constexpr int BM = 32, BN = 32 * TN16, BK = 32, NWAVES = 4, THREADS = 256;
    static constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK;
    static constexpr int SMEM_BYTES = NBUF * (A_FLOATS + B_FLOATS) * 4;
    static constexpr int NA = BM / 8, NB = BN / 8, NI = (NA + NB) / NWAVES;       // 1-KiB pieces: 8 rows of 128 bytes
    static_assert(NA % NWAVES == 0 && NB % NWAVES == 0, "pieces must divide evenly over the waves");
};

#if defined(__HIP_DEVICE_COMPILE__)
// tile index of workgroup `bid` of `nwg`: each XCD (bid % 8) walks a contiguous range
__device__ __forceinline__ int xcd_swz(int bid, int nwg) {
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
}

template <int TN16_, int NBUF_, bool LATE>
__device__ __forceinline__ void s16_tile(const GemmArgs& p, float* smem, int swz) {
    using C_ = S16Cfg<TN16_, NBUF_>;
    constexpr int BM = C_::BM, BN = C_::BN, BK = C_::BK, TN = C_::TN16, NBUF = C_::NBUF, NI = C_::NI, UA = C_::NA / C_::NWAVES;
    float* As = smem;                                   // [NBUF][BM*BK]
    float* Bs = smem + NBUF * C_::A_FLOATS;             // [NBUF][BN*BK]
    const int m0 = (swz / p.nt) * BM;
    const int n0 = (swz % p.nt) * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;

    int voff[NI], dst[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int pq = wave + C_::NWAVES * u;
        const int prow = lane >> 3, pc = lane & 7;
        const bool isa = u < UA;
        const int trow = 8 * (isa ? pq : pq - C_::NA) + prow;
        const int gch = pc ^ ((trow >> 1) & 7);
        if (isa) {
            const int rel = min(trow, p.M - 1 - m0);                            // rows past M re-read the last row (never stored)
            voff[u] = (rel * p.lda + 4 * gch) * 4;
            dst[u] = 8 * pq * BK;
        } else {
            const int rel = min(trow, p.N - 1 - n0);
            voff[u] = (rel * p.ldw + 4 * gch) * 4;
            dst[u] = NBUF * C_::A_FLOATS + 8 * (pq - C_::NA) * BK;
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + (size_t)m0 * p.lda), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W + (size_t)n0 * p.ldw), 0, 0xffffffff, 0x00020000);
    const int nkt = p.K / BK;
    auto stage = [&](int buf, int kt) {
        const int koff = min(kt, nkt - 1) * (BK * 4);
#pragma unroll
        for (int u = 0; u < NI; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(u < UA ? rsA : rsW, (lptr_t)(smem + dst[u] + buf * (u < UA ? C_::A_FLOATS : C_::B_FLOATS)), 16, voff[u], koff, 0, 0);
    };

    // accumulators start as the bias; the residual / PE quads are requested here too (the oldest loads: the loop's counted waits cover them)
    const int row = m0 + wm * 16 + r;
    const int colw = n0 + wn * (16 * TN) + 4 * q;
    f32x4 acc[TN], rv[LATE ? TN : 1];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = colw + 16 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (col < p.N && p.bias) v = *reinterpret_cast<const f32x4*>(p.bias + col);
        acc[j] = v;
        if constexpr (LATE) {
            const int er = p.epilogue == MMDM_EPI_BIAS_PE ? row % p.period : row;
            const int rr = row < p.M ? er : 0, cc = col < p.N ? col : 0;         // past the edge: a valid address, never stored
            rv[j] = *reinterpret_cast<const f32x4*>(p.extra + (size_t)rr * p.ld_extra + cc);
        }
    }
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) stage(t, t);

    const int sw = (r >> 1) & 7;                       // every tile row of this lane is r mod 16
    const int a_row = (wm * 16 + r) * BK;
    const int b_row = (wn * (16 * TN) + r) * BK;
    const bool hi = (q >> 1) != 0;
    int cur = 0, stg = NBUF - 1;
    for (int kt = 0; kt < nkt; ++kt) {
        wait_vm<(NBUF - 2) * NI>();                    // tile kt has landed (in-order retirement; NI requests per stage, surplus ones included)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // ... for every wave, and every wave is done with stage (kt - 1) % NBUF
        stage(stg, kt + NBUF - 1);
        // k-slot q of a group's first MFMA: 8g + {0, 4, 1, 5}[q] = element (q >> 1) of chunk 2g + (q & 1); of its second: element 2 + (q >> 1).
        // One 16-byte fragment read per operand row and group, two selects.  (Two dwords by ds_read2_b32 -- half the LDS bytes -- and all of a
        // step's reads ahead of its MFMAs were measured: 10-25 % slower on every shape; so was a software-pipelined ring -- a step's MFMAs fed from
        // registers read during the previous step, requests and reads interleaved behind them: 1196 x 1024 x 1024 36.3 -> 45.9 us alone, 30.1 -> 32.7
        // in the mixed launch, the B = 1 step 5.46 -> 5.76 ms.  tools/gemm_s16_bench.py.)
        const float* Ac = As + cur * C_::A_FLOATS + a_row;
        const float* Bc = Bs + cur * C_::B_FLOATS + b_row;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int cg = 4 * ((2 * g + (q & 1)) ^ sw);
            const f32x4 va = *reinterpret_cast<const f32x4*>(Ac + cg);
            f32x4 vb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) vb[j] = *reinterpret_cast<const f32x4*>(Bc + j * 16 * BK + cg);
            const float a0 = hi ? va[1] : va[0], a1 = hi ? va[3] : va[2];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi ? vb[j][1] : vb[j][0], a0, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi ? vb[j][3] : vb[j][2], a1, acc[j], 0, 0, 0);
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        stg = stg + 1 == NBUF ? 0 : stg + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus requests of the last steps
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[j]));

    const int rows_here = min(p.M - m0, BM);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C + (size_t)m0 * p.ldc, 0, rows_here * p.ldc * 4, 0x00020000);
    const int voffC = (wm * 16 + r) * p.ldc * 4 + (colw - n0) * 4 + n0 * 4;
    auto finish = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = acc[j][c];
                if constexpr (LATE) t += rv[j][c];
                if constexpr (ACT == MMDM_EPI_BIAS_GELU) t = gelu_erf(t);
                else if constexpr (ACT == MMDM_EPI_BIAS_SILU) t = silu(t);
                else if constexpr (ACT == MMDM_EPI_BIAS_QUICKGELU) t = quick_gelu(t);
                else if constexpr (ACT == MMDM_EPI_BIAS_SIGMOID) t = sigmoidf(t);
                v[c] = t;
            }
            if (colw + 16 * j < p.N) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, voffC + 64 * j, 0, 0);
        }
    };
    if constexpr (LATE) {
        finish(std::integral_constant<int, MMDM_EPI_BIAS>{});
    } else {
        switch (p.epilogue) {
            case MMDM_EPI_BIAS_GELU: finish(std::integral_constant<int, MMDM_EPI_BIAS_GELU>{}); break;
            case MMDM_EPI_BIAS_SILU: finish(std::integral_constant<int, MMDM_EPI_BIAS_SILU>{}); break;
            case MMDM_EPI_BIAS_QUICKGELU: finish(std::integral_constant<int, MMDM_EPI_BIAS_QUICKGELU>{}); break;
            case MMDM_EPI_BIAS_SIGMOID: finish(std::integral_constant<int, MMDM_EPI_BIAS_SIGMOID>{}); break;
            default: finish(std::integral_constant<int, MMDM_EPI_BIAS>{}); break;
        }
    }
}
#endif

template <int TN16_, int NBUF_, bool LATE>
__global__ __launch_bounds__(256) void gemm_s16_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    s16_tile<TN16_, NBUF_, LATE>(p, smem, xcd_swz(blockIdx.x, p.mt * p.nt));
#endif
}

// One launch, two tilings (the reference's own call shape, B = 1: M = 1196 rows).  With 64 x 64 tiles of 32 x 32 chains the rows that fill WHOLE
// rounds of the 256 CUs -- 1024 of them at N = 1024 / 2048 / 3072 -- cost one chain length per round; the 172 rows behind them cost a whole further
// round on a fifth of the chip.  Here workgroups [0, n_main) run the production 64 x 64 tile on the whole-round rows and the
// workgroups after them the 16 x 16-chain tile on the remaining rows: quarter-length chains that the dispatcher places beside the main tiles
// as soon as those are resident (both forms take 32 KB of LDS and 256 threads), so the launch lasts the whole rounds plus what the matrix
// pipes still owe the remainder.  Same chains, same order: bit-identical to either kernel alone.
struct MixArgs { GemmArgs main, rem; int n_main; };

template <bool VEPI>
__global__ __launch_bounds__(256) void gemm_mix_kernel(MixArgs q) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    if (bid < q.n_main) gemm_tile<21, 21, 16, 4, VEPI, 1, false>(q.main, smem, xcd_swz(bid, q.n_main), bid);
    else s16_tile<1, 4, VEPI>(q.rem, smem, xcd_swz(bid - q.n_main, q.rem.mt * q.rem.nt));
#endif
}

// what the 16-byte accesses of gemm_s16_kernel need (every epilogue: bias quads, result quads; residual / PE rows)
inline bool s16_ok(const GemmArgs& a) {
    const bool ext = a.epilogue == MMDM_EPI_BIAS_RESID || a.epilogue == MMDM_EPI_BIAS_PE;
    return (a.N & 3) == 0 && (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && (a.K & 31) == 0 && a.Kw == a.K &&
           (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
           (!ext || ((a.ld_extra & 3) == 0 && (reinterpret_cast<uintptr_t>(a.extra) & 15) == 0));
}

template <int TN16_, int NBUF_>
int launch_s16(GemmArgs a, hipStream_t st) {
    using C_ = S16Cfg<TN16_, NBUF_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    const bool late = a.epilogue == MMDM_EPI_BIAS_RESID || a.epilogue == MMDM_EPI_BIAS_PE;
    mmdm_note_gemm("gemm_s16<%d,%d,%s>", TN16_, NBUF_, late ? "late" : "plain");
    const dim3 grid(a.mt * a.nt), block(C_::THREADS);
    if (late) hipLaunchKernelGGL((gemm_s16_kernel<TN16_, NBUF_, true>), grid, block, C_::SMEM_BYTES, st, a);
    else hipLaunchKernelGGL((gemm_s16_kernel<TN16_, NBUF_, false>), grid, block, C_::SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_s16");
}

// rows [0, M1) on 64 x 64 tiles, rows [M1, M) on the 16 x 16-chain tiles, one launch (a.epilogue != PE: the row index restarts in the remainder)
int launch_mix(const GemmArgs& a, int M1, hipStream_t st) {
    using CM = GCfg<21, 21, 16, 4>;
    using CR = S16Cfg<1, 4>;
    static_assert(CM::SMEM_BYTES == CR::SMEM_BYTES && CM::THREADS == CR::THREADS, "the two tile forms share one launch configuration");
    MixArgs q;
    q.main = a; q.rem = a;
    q.main.M = M1;
    q.rem.M = a.M - M1;
    q.rem.A = a.A + (size_t)M1 * a.lda;
    q.rem.C = a.C + (size_t)M1 * a.ldc;
    if (a.extra) q.rem.extra = a.extra + (size_t)M1 * a.ld_extra;
    q.main.mt = (M1 + CM::BM - 1) / CM::BM; q.main.nt = (a.N + CM::BN - 1) / CM::BN;
    q.rem.mt = (q.rem.M + CR::BM - 1) / CR::BM; q.rem.nt = (a.N + CR::BN - 1) / CR::BN;
    q.main.ablate = q.rem.ablate = 0; q.main.stamps = q.rem.stamps = nullptr;
    q.n_main = q.main.mt * q.main.nt;
    const bool ext = a.epilogue == MMDM_EPI_BIAS_RESID;
    mmdm_note_gemm("gemm_mix<%s,%d+%d>", ext ? "vepi" : "scalar", q.n_main, q.rem.mt * q.rem.nt);
    const dim3 grid(q.n_main + q.rem.mt * q.rem.nt), block(CM::THREADS);
    if (ext) hipLaunchKernelGGL((gemm_mix_kernel<true>), grid, block, CM::SMEM_BYTES, st, q);
    else hipLaunchKernelGGL((gemm_mix_kernel<false>), grid, block, CM::SMEM_BYTES, st, q);
    return mmdm_check_launch("gemm_mix");
}

int set_attr_mix() {
    const void* fns[2] = {reinterpret_cast<const void*>(&gemm_mix_kernel<true>), reinterpret_cast<const void*>(&gemm_mix_kernel<false>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, GCfg<21, 21, 16, 4>::SMEM_BYTES);
        if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_mix): %s", hipGetErrorString(e));
    }
    return MMDM_OK;
}

template <int TN16_, int NBUF_>
int set_attr_s16() {
    constexpr int bytes = S16Cfg<TN16_, NBUF_>::SMEM_BYTES;
    const void* fns[2] = {reinterpret_cast<const void*>(&gemm_s16_kernel<TN16_, NBUF_, true>), reinterpret_cast<const void*>(&gemm_s16_kernel<TN16_, NBUF_, false>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_s16): %s", hipGetErrorString(e));
    }
    return MMDM_OK;
}

inline bool vepi_ok(const GemmArgs& a) {
    const bool ext = a.epilogue == MMDM_EPI_BIAS_RESID || a.epilogue == MMDM_EPI_BIAS_PE;
    return (a.N & 3) == 0 && (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
           (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
           (!ext || ((a.ld_extra & 3) == 0 && (reinterpret_cast<uintptr_t>(a.extra) & 15) == 0));
}

template <int TM_, int TN_, int BK_ = 16, int NBUF_ = 2, int MINW_ = 1, int PIPE_ = 0>
int launch_glds(GemmArgs a, hipStream_t st) {
    using C_ = GCfg<TM_, TN_, BK_, NBUF_>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    // measured (tools/gemm_bench.py): the 16-byte epilogue pays where the accumulators are initialised from memory (+10 % on
    // the K = 1024 residual GEMMs) and is neutral-to-slightly-negative for bias/GELU-only epilogues
    const bool ext = a.epilogue == MMDM_EPI_BIAS_RESID || a.epilogue == MMDM_EPI_BIAS_PE;
    const bool vepi = ext && vepi_ok(a) && !(a.ablate & 16);
    const dim3 grid(a.mt * a.nt), block(C_::THREADS);
    mmdm_note_gemm("%s<%d,%d,%d,%d,%s>", PIPE_ ? "gemm_pipe" : "gemm_glds", TM_, TN_, BK_, NBUF_, vepi ? "vepi" : "scalar");
    if ((a.ablate & ~16) || a.stamps) {              // a tool asked for ablation bits / stamps: the diagnostic instantiation
        if (vepi) hipLaunchKernelGGL((gemm_glds_kernel<TM_, TN_, BK_, NBUF_, true, MINW_, PIPE_, true>), grid, block, C_::SMEM_BYTES, st, a);
        else hipLaunchKernelGGL((gemm_glds_kernel<TM_, TN_, BK_, NBUF_, false, MINW_, PIPE_, true>), grid, block, C_::SMEM_BYTES, st, a);
    } else if (vepi)
        hipLaunchKernelGGL((gemm_glds_kernel<TM_, TN_, BK_, NBUF_, true, MINW_, PIPE_, false>), grid, block, C_::SMEM_BYTES, st, a);
    else
        hipLaunchKernelGGL((gemm_glds_kernel<TM_, TN_, BK_, NBUF_, false, MINW_, PIPE_, false>), grid, block, C_::SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_glds");
}

template <int TM_, int TN_, int BK_ = 16, int NBUF_ = 2, int MINW_ = 1, int PIPE_ = 0>
int set_attr_glds() {
    constexpr int bytes = GCfg<TM_, TN_, BK_, NBUF_>::SMEM_BYTES;
    const void* fns[4] = {reinterpret_cast<const void*>(&gemm_glds_kernel<TM_, TN_, BK_, NBUF_, true, MINW_, PIPE_, false>),
                          reinterpret_cast<const void*>(&gemm_glds_kernel<TM_, TN_, BK_, NBUF_, false, MINW_, PIPE_, false>),
                          reinterpret_cast<const void*>(&gemm_glds_kernel<TM_, TN_, BK_, NBUF_, true, MINW_, PIPE_, true>),
                          reinterpret_cast<const void*>(&gemm_glds_kernel<TM_, TN_, BK_, NBUF_, false, MINW_, PIPE_, true>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm_glds): %s", hipGetErrorString(e));
    }
    return MMDM_OK;
}

template <int TM, int TN, int BK, int AVEC, int WVEC, bool FULL>
int set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<TM, TN, BK, AVEC, WVEC, FULL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<TM, TN, BK>::SMEM_BYTES);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
    return MMDM_OK;
}

template <int TM, int TN, int BK, int AVEC, int WVEC, bool FULL>
int launch(GemmArgs a, hipStream_t st) {
    using C_ = Cfg<TM, TN, BK>;
    a.mt = (a.M + C_::BM - 1) / C_::BM;
    a.nt = (a.N + C_::BN - 1) / C_::BN;
    mmdm_note_gemm("gemm_f32<%d,%d,%d>", TM, TN, BK);
    hipLaunchKernelGGL((gemm_f32_kernel<TM, TN, BK, AVEC, WVEC, FULL>), dim3(a.mt * a.nt), dim3(C_::THREADS), C_::SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_f32");
}

// tile variants (selected per call; MMDM_GEMM_CFG overrides for benchmarking)
template <int TM, int TN, int BK>
int launch_cfg(const GemmArgs& a, bool av, bool wv, hipStream_t st) {
    using C_ = Cfg<TM, TN, BK>;
    const bool full = av && wv && (a.M % C_::BM == 0) && (a.N % C_::BN == 0) && (a.K % C_::BK == 0) && a.Kw == a.K;
    if (full) return launch<TM, TN, BK, 4, 4, true>(a, st);
    if (av && wv) return launch<TM, TN, BK, 4, 4, false>(a, st);
    if (!av && wv) return launch<TM, TN, BK, 1, 4, false>(a, st);
    if (av && !wv) return launch<TM, TN, BK, 4, 1, false>(a, st);
    return launch<TM, TN, BK, 1, 1, false>(a, st);
}

template <int TM, int TN, int BK>
int set_attr_cfg() {
    int rc;
    if ((rc = set_attr<TM, TN, BK, 4, 4, true>())) return rc;
    if ((rc = set_attr<TM, TN, BK, 4, 4, false>())) return rc;
    if ((rc = set_attr<TM, TN, BK, 1, 4, false>())) return rc;
    if ((rc = set_attr<TM, TN, BK, 4, 1, false>())) return rc;
    return set_attr<TM, TN, BK, 1, 1, false>();
}

inline bool vec_ok(const float* p, int ld, int K) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (K & 3) == 0;
}

}  // namespace

int g_gemm_cfg = -1;
int g_gemm_tail = -1;       // row split of the fractional last round: t > 0: split when the fractional round holds <= t/10 of the resident slots; 0 off;
                            // -1 (default): what the caller's handle asked for through mmdm_gemm_set_tail (one-stream samplers: 10, two-stream: 0)
static thread_local int t_gemm_tail = 0;
void mmdm_gemm_set_tail(int t) { t_gemm_tail = t; }
int mmdm_gemm_get_tail(void) { return t_gemm_tail; }
int g_gemm_ablate = 0;
#ifndef MMDM_S16_DEFAULT
#define MMDM_S16_DEFAULT -1
#endif
int g_gemm_s16 = MMDM_S16_DEFAULT;       // mmdm_diag_set "gemm_s16": -1 automatic, 0 off, 13 / 23 / 14 / 24: force <TN16, stages> wherever the kernel covers the call
unsigned long long* g_gemm_stamps = nullptr;

int mmdm_gemm_init(void) {
    int rc;
    if ((rc = set_attr_cfg<22, 22, 32>())) return rc;
    if ((rc = set_attr_cfg<22, 22, 16>())) return rc;
    if ((rc = set_attr_cfg<22, 12, 16>())) return rc;
    if ((rc = set_attr_cfg<22, 22, 8>())) return rc;
    if ((rc = set_attr_cfg<42, 22, 16>())) return rc;
    if ((rc = set_attr_cfg<22, 42, 16>())) return rc;
    if ((rc = set_attr_glds<42, 22>())) return rc;
    if ((rc = set_attr_glds<22, 22>())) return rc;
    if ((rc = set_attr_glds<22, 21>())) return rc;
    if ((rc = set_attr_glds<22, 22, 16, 4, 1, 1>())) return rc;
    if ((rc = set_attr_glds<42, 22, 16, 3, 1, 1>())) return rc;
    if ((rc = set_attr_glds<22, 21, 16, 3, 1, 1>())) return rc;
    if ((rc = set_attr_glds<22, 22, 16, 5, 1, 1>())) return rc;
    if ((rc = set_attr_glds<22, 21, 16, 4, 1, 1>())) return rc;
    if ((rc = set_attr_glds<22, 21, 16, 5, 1, 1>())) return rc;
    if ((rc = set_attr_glds<21, 21, 16, 4, 1, 1>())) return rc;
    if ((rc = set_attr_glds<22, 22, 16, 5, 1, 2>())) return rc;
    if ((rc = set_attr_glds<22, 22, 16, 5, 1, 3>())) return rc;
    if ((rc = set_attr_s16<1, 3>())) return rc;
    if ((rc = set_attr_s16<2, 3>())) return rc;
    if ((rc = set_attr_s16<1, 4>())) return rc;
    if ((rc = set_attr_s16<2, 4>())) return rc;
    if ((rc = set_attr_mix())) return rc;
    return MMDM_OK;
}

// diagnostics of this translation unit (mmdm_diag_set, include/mmdm.h section 4): tile configuration override (-1 = automatic), timing
// ablation bits, row-split rule override, in-kernel stamp buffer.  Process-global, for tools/ only.
bool mmdm_diag_gemm_f32(const char* key, long long v) {
    if (!strcmp(key, "gemm_cfg")) g_gemm_cfg = (int)v;
    else if (!strcmp(key, "gemm_ablate")) g_gemm_ablate = (int)v;
    else if (!strcmp(key, "gemm_tail")) g_gemm_tail = (int)v;
    else if (!strcmp(key, "gemm_s16")) g_gemm_s16 = (int)v;
    else if (!strcmp(key, "gemm_stamps")) g_gemm_stamps = reinterpret_cast<unsigned long long*>((uintptr_t)v);
    else return false;
    return true;
}

extern "C" int mmdm_linear_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                               int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_f32_ex(A, lda, W, ldw, K, bias, C, ldc, M, N, K, epilogue, extra, ld_extra, period, stream);
}

// Kw: number of readable columns in each W row (>= K, columns K..Kw-1 must be zero) -- lets a K = 262 weight that the
// handle stores zero-padded to 264 columns be fetched with 16-byte loads.
int mmdm_linear_f32_ex(const float* A, int lda, const float* W, int ldw, int Kw, const float* bias, float* C, int ldc,
                       int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    mmdm_note_gemm_reset();
    if (M == 0 || N == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    if (!A || !W || !C || M < 0 || N < 0 || K <= 0 || lda < K || ldw < Kw || Kw < K || ldc < N)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: bad shape M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
    if (epilogue < MMDM_EPI_BIAS || epilogue > MMDM_EPI_BIAS_SIGMOID)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: unknown epilogue %d", epilogue);
    if ((epilogue == MMDM_EPI_BIAS_RESID || epilogue == MMDM_EPI_BIAS_PE) && (!extra || ld_extra < N))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: epilogue %d needs `extra` with ld >= N", epilogue);
    if (epilogue == MMDM_EPI_BIAS_PE && period <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: PE epilogue needs period > 0");
    GemmArgs a;
    a.A = A; a.W = W; a.bias = bias; a.C = C; a.extra = extra;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ld_extra = ld_extra;
    a.M = M; a.N = N; a.K = K; a.Kw = Kw; a.epilogue = epilogue; a.period = period > 0 ? period : 1;
    a.mt = a.nt = 0;
    a.ablate = g_gemm_ablate;
    a.stamps = g_gemm_stamps;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool av = vec_ok(A, lda, K), wv = vec_ok(W, ldw, Kw);
    const bool glds_ok = av && wv && (K % 16 == 0) && Kw == K;
    // production choice: LDS-DMA kernel, 128x128 tile / 4 waves (5 workgroups per CU) whenever the operands allow it
    const int tail = g_gemm_tail >= 0 ? g_gemm_tail : t_gemm_tail;
    switch (g_gemm_cfg) {
        case 10: if (glds_ok) return launch_glds<42, 22>(a, st); break;
        case -1:
            if (glds_ok && K >= 96 && tail && epilogue != MMDM_EPI_BIAS_PE) {
                // Row split ("tile list" of two entries) for callers that run ONE stream of kernels: M = 12 544 (configs[1]) gives 2352 / 3136 /
                // 784 tiles for 512 (128x128, two per CU) or 768 (128x64, three per CU) resident slots = 4.6 / 4.1 / 1.5 rounds, and the
                // fractional round costs a whole tile lifetime on a partly empty chip.  Rows [0, M1) -- as many whole row tiles as fit into
                // complete rounds -- go to the large tile; the remaining rows to a tile of half / a quarter of the area, whose single
                // partial round is as short.  Measured: configs[1] 15.35 -> 14.45 ms/step; QKV at M = 19 200 stand-alone 894 -> 857 us.
                // The two-stream MixerMDM step does NOT use it (59.2 vs 59.5 ms/step: the other stream's kernels already fill a tail).
                // Every pipelined instantiation accumulates an output element in the same order, so the split does not change a bit
                // (tests: bitwise batch independence, production tiles vs float64).
                const bool narrow = N <= 512 || K <= 512 || N == 2048;
                const long slots = narrow ? 768 : 512, nt = (N + (narrow ? 63 : 127)) / (narrow ? 64 : 128), mt = (M + 127) / 128;
                const long tiles = mt * nt, full = tiles / slots, rem = tiles - full * slots;
                const long mt1 = full * slots / nt;
                if ((long)((M + 127) / 128) * ((N + 63) / 64) >= 512 && full >= 1 && mt1 >= 1 && mt1 < mt && rem * 10 <= slots * tail) {
                    const int M1 = (int)mt1 * 128;
                    GemmArgs b = a;
                    a.M = M1;
                    b.M = M - M1;
                    b.A = A + (size_t)M1 * lda;
                    b.C = C + (size_t)M1 * ldc;
                    if (extra) b.extra = extra + (size_t)M1 * ld_extra;
                    int rc = narrow ? launch_glds<22, 21, 16, 4, 1, 1>(a, st) : launch_glds<22, 22, 16, 5, 1, 1>(a, st);
                    if (rc) return rc;
                    // remainder: the largest tile that still gives every CU a workgroup
                    const long r128 = (long)((b.M + 127) / 128) * ((N + 63) / 64);
                    if (!narrow && r128 >= 256) return launch_glds<22, 21, 16, 4, 1, 1>(b, st);
                    return launch_glds<21, 21, 16, 4, 1, 1>(b, st);
                }
            }
            if (glds_ok && K >= 96) {
                // Production: the software-pipelined loop (gemm_pipe).  Tile / stage choice per shape, measured at M = 19 200 with
                // tools/gemm_bench.py (TFLOP/s; the 2-stage kernels of round 1 reached 100-121 on the same shapes):
                //   N = 3072, K = 1024 (QKV):        128x128, 5 stages 128      N = 2048, K = 1024 (FFN up, K|V): 128x64, 4 stages 119-122
                //   N = 1024, K = 1024 / 2048:       128x128, 5 stages 124 / 128   mixer (N or K <= 512):          128x64, 4 stages 107-121
                // Few tiles (small batches): 64x64 tiles so that every CU has work.  All instantiations accumulate each output element in
                // the same order (bias + residual first, then k ascending), so a row's result does not depend on the tile that produced it.
                const long t64 = (long)((M + 127) / 128) * ((N + 63) / 64);
                if (g_gemm_s16 > 0 && s16_ok(a)) {
                    switch (g_gemm_s16) {
                        case 13: return launch_s16<1, 3>(a, st);
                        case 14: return launch_s16<1, 4>(a, st);
                        case 24: return launch_s16<2, 4>(a, st);
                        default: return launch_s16<2, 3>(a, st);
                    }
                }
                // Launches whose 64 x 64 grid leaves half of the CUs idle (configs[0]'s M = 240 against N <= 2048, conditioning / time-embedding rows):
                // quarter-length chains on 16 x 16 blocks, four times the workgroups, bit-identical (gemm_s16_kernel).  Not the weight-streaming
                // projections (M <= 64 against N >= 4096: below).
                if (g_gemm_s16 < 0 && (long)((M + 63) / 64) * ((N + 63) / 64) <= 128 && !(M <= 64 && N >= 4096) && s16_ok(a)) return launch_s16<1, 4>(a, st);
                // 64 x 64 tiles with a fractional last round (the reference's B = 1 call: 19 row tiles x N / 64 = 1.2 / 2.4 / 3.6 rounds of 256): the rows of
                // the whole rounds on those tiles, the rest as quarter-length chains beside them, in one launch (gemm_mix_kernel)
                if (t64 < 512 && g_gemm_s16 < 0 && epilogue != MMDM_EPI_BIAS_PE && s16_ok(a) && vepi_ok(a)) {
                    // measured (tools/gemm_s16_bench.py, M = 1196; us, 64 x 64 launch -> mixed): N = 1024: 37.6 -> 30.1 (K = 2048: 63.5 -> 50.1), N = 2048: 51.2 -> 47.9,
                    // N = 3072 (three whole rounds): 79.8 -> 82.7, N = 1536 at K = 512 (more remainder than main tiles): 19.5 -> 23.7 -- the remainder tiles
                    // cost about twice their matrix-pipe time, so the launch is mixed for one or two whole rounds and a remainder no larger than the main part
                    const long nt64 = (N + 63) / 64, tiles = (long)((M + 63) / 64) * nt64, rounds = tiles / 256, mt1 = rounds * 256 / nt64;
                    const long rem_wgs = mt1 * 64 < M ? (long)((M - mt1 * 64 + 31) / 32) * ((N + 31) / 32) : 0;
                    if (rounds >= 1 && rounds <= 2 && mt1 >= 1 && rem_wgs > 0 && rem_wgs <= mt1 * nt64 && tiles % 256 != 0) return launch_mix(a, (int)mt1 * 64, st);
                }
                if (t64 < 512) return launch_glds<21, 21, 16, 4, 1, 1>(a, st);
                // Skinny M against a wide weight matrix -- the packed AdaLN projections of a step: M = 2B .. 4B conditioning rows, N = L x n_ada x 2D =
                // 32 768 / 49 152 columns, 134 / 201 MB of fp32 weights read ONCE per step: a weight-streaming launch, not a matrix-pipe one.  The 128 x 128
                // tile gives it 384 workgroups of 80 KB (two per CU) and multiplies padding rows: 158 us for M = 64, N = 49 152 with W cold (1.3 TB/s);
                // 64 x 64 tiles (768 workgroups, 32 KB: five per CU, more bytes in flight) 79 us; 128 x 64 for 64 < M <= 256: 134 -> 110 us (M = 128),
                // 258 -> 207 (M = 256).  Bit-identical like every tile choice (round 6; tools: a probe with W evicted between calls).
                if (M <= 64 && N >= 4096) return launch_glds<21, 21, 16, 4, 1, 1>(a, st);
                if (M <= 256 && N >= 4096) return launch_glds<22, 21, 16, 4, 1, 1>(a, st);
                if (N <= 512 || K <= 512 || N == 2048) return launch_glds<22, 21, 16, 4, 1, 1>(a, st);
                return launch_glds<22, 22, 16, 5, 1, 1>(a, st);
            }
            if (glds_ok) {
                const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
                return tiles < 1000 ? launch_glds<22, 21>(a, st) : launch_glds<22, 22>(a, st);
            }
            break;
        case 11: if (glds_ok) return launch_glds<22, 22>(a, st); break;
        case 17: if (glds_ok) return launch_glds<22, 21>(a, st); break;     // 128 x 64 tile, 4 waves (64 x 32 per wave)
        case 31: if (glds_ok && K >= 64) return launch_glds<22, 22, 16, 4, 1, 1>(a, st); break;   // ... 4 stages (64 KB: 2 per CU)
        case 32: if (glds_ok && K >= 64) return launch_glds<42, 22, 16, 3, 1, 1>(a, st); break;   // ... 256 x 128 / 8 waves, 3 stages (72 KB: 2 per CU)
        case 33: if (glds_ok && K >= 64) return launch_glds<22, 21, 16, 3, 1, 1>(a, st); break;   // ... 128 x 64 / 4 waves
        case 34: if (glds_ok && K >= 80) return launch_glds<22, 22, 16, 5, 1, 1>(a, st); break;   // ... 128 x 128, 5 stages (80 KB: 2 per CU)
        case 36: if (glds_ok && K >= 64) return launch_glds<22, 21, 16, 4, 1, 1>(a, st); break;   // ... 128 x 64, 4 stages (48 KB: 3 per CU)
        case 38: if (glds_ok && K >= 80) return launch_glds<22, 21, 16, 5, 1, 1>(a, st); break;   // ... 128 x 64, 5 stages (60 KB: 2 per CU)
        case 41: if (glds_ok && K >= 64) return launch_glds<21, 21, 16, 4, 1, 1>(a, st); break;   // ... 64 x 64 / 4 waves of 32 x 32
        case 60: if (glds_ok && K >= 96) return launch_glds<22, 22, 16, 5, 1, 2>(a, st); break;   // ablation: no LDS-DMA in the main loop
        case 61: if (glds_ok && K >= 96) return launch_glds<22, 22, 16, 5, 1, 3>(a, st); break;   // ablation: no barrier
        default: break;
    }
    switch (g_gemm_cfg) {
        case 0: return launch_cfg<22, 22, 32>(a, av, wv, st);
        case 2: return launch_cfg<22, 12, 16>(a, av, wv, st);
        case 3: return launch_cfg<22, 22, 8>(a, av, wv, st);
        case 5: return launch_cfg<22, 42, 16>(a, av, wv, st);
        case 4: return launch_cfg<42, 22, 16>(a, av, wv, st);
        default: return launch_cfg<22, 22, 16>(a, av, wv, st);
    }
}
