// Exact-fp32 multi-head attention core with add_zero_attn for gfx950 (v_mfma_f32_16x16x4_f32, flash-style).
//
// Replaces the scaled-dot-product attention inside nn.MultiheadAttention(add_zero_attn=True) as used by
// VanillaSelfAttention / VanillaCrossAttention (reference: src/models/utils/layers.py:33-44, 74-87).
//   out[s,q,h,:] = softmax_{keys + one zero key}( Q[s,q,h,:].K[s',k,h,:] / sqrt(dh) ) V[s',k,h,:],  s' = (s+shift) % nseq
// The extra key has logit 0 and value 0, so it is folded in as the INITIAL online-softmax state (m=0, l=1, O=0).
//
// Structure: one 256-thread workgroup = 4 waves = 64 queries of one (sequence, head); each wave owns 16 queries.
// Keys/values stream through LDS in stages of 16 keys, double-buffered by LDS-DMA (global_load_lds_dwordx4: the next stage
// is in flight while the current one is consumed; unpadded rows, XOR-swizzled on the DMA source address and on the read).
// QK^T is computed swapped (S^T = K Q^T) so that each lane holds
// four keys of ONE query column (C/D map: col = lane&15 = query, row = 4*(lane>>4)+reg = key): the row max / row sum
// are 4 local values + two xor-shuffles (16, 32), and the exponentiated tile is already the A operand of the
// P.V MFMA (k index = lane group) with no cross-lane movement or LDS round trip.  The d (reduction) order of
// QK^T is permuted identically for both operands so that one 16-byte LDS read feeds four MFMAs.
#include <hip/hip_runtime.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include "kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define EXP2(x) __builtin_amdgcn_exp2f(x)
// hipcc (ROCm 7.2) pads the MFMA -> VALU read-after-write hazard only inside a basic block: when a (wave-uniform) branch
// follows the last v_mfma of a chain and the branch target starts with a VALU read of the accumulator, no wait states are
// emitted on the taken path and the VALU can read the accumulator before the MFMA has written it (seen here as a run-to-run
// different running max once two waves shared a SIMD).  MFMA_SETTLE(x) is an opaque asm statement that takes the accumulator
// as a read-write operand -- so the compiler keeps it behind the MFMAs that produce x and ahead of every consumer -- and
// supplies the wait states itself (24 >= the 18 a 16-pass MFMA needs).
#define MFMA_SETTLE(x) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(x))


// Output-column ownership of the P.V half: lane (lq = lane & 15) owns the NJ = DH/16 CONTIGUOUS columns [NJ*lq, NJ*lq + NJ) of its head (tile j of
// the 16x16 MFMA output is column NJ*lq + j for that lane).  The B operand of a P.V MFMA is then V[key][NJ*lq .. +NJ): one or two ds_read_b128 per
// key row instead of NJ scalar reads, and the result is stored with 16-byte accesses.  For DH = 128 (32 bytes per lane) lanes lq >= 8 take their
// two 16-byte halves in the opposite order (tile j <-> column NJ*lq + (j ^ 4)), which makes every ds_read_b128 lane group (MI355X_MICROARCH.md:
// {0-3,12-15,20-27}, ...) hit 64 distinct banks with an unswizzled V image; the permutation is undone by the store addresses.
template <int DH>
__device__ __forceinline__ void load_v_row(const float* __restrict__ vrow_base, int lq, float (&vb)[DH / 16]) {
    if constexpr (DH == 128) {
        const int h0 = lq >= 8 ? 4 : 0;
        const f32x4 a = *reinterpret_cast<const f32x4*>(vrow_base + 8 * lq + h0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(vrow_base + 8 * lq + (h0 ^ 4));
        vb[0] = a[0]; vb[1] = a[1]; vb[2] = a[2]; vb[3] = a[3];
        vb[4] = b[0]; vb[5] = b[1]; vb[6] = b[2]; vb[7] = b[3];
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(vrow_base + 4 * lq);
        vb[0] = a[0]; vb[1] = a[1]; vb[2] = a[2]; vb[3] = a[3];
    }
}


constexpr int KC = 16;   // keys per LDS stage (attn_qkp_kernel)
constexpr int KCF = 16;  // keys per LDS stage of attn_mfma_kernel (32 measured: dh = 128 271 us vs 258, dh = 64 129 vs 131 -- 300 keys pad to 320 and occupancy halves)
constexpr int NST = 2;   // LDS stages of attn_mfma_kernel: chunk ci + NST - 1 is requested while chunk ci is consumed (counted vmcnt, raw barrier)
constexpr int QW = 16;   // queries per wave
constexpr int QB = 64;   // queries per workgroup

struct AttnArgs {
    const float* Q; const float* K; const float* V; float* O;
    int ldq, ldk, ldv, ldo;
    int nseq, Tq, Tk, H, shift, qtiles, pairs_per_xcd;
    float scale, scale2;
    int out_bf16;      // 1: O written as bf16 (feeds the bf16 out-projection GEMM); 2: as the two fp16 planes of the fp32-split mode (kernels.h mmdm_split2; plane stride nseq*Tq*ldo)
    unsigned long long* stamps;   // diagnostic launches only (tools/attn_timeline.py): per workgroup {entry, loop start, loop end, kernel end, placement, qk, softmax, pv} in 100 MHz ticks
    int ablate;        // timing experiments only (tools/attn_bench.py, mmdm_diag_set "attn_ablate"); 0 in production
    int flags;         // MMDM_ATTN_NO_ZERO_KEY: plain softmax (nn.MultiheadAttention default); MMDM_ATTN_CAUSAL: key <= query only
    int dh;            // real head width (<= the kernel's DH, multiple of 4): heads narrower than the template (the 96-wide heads of the
                       // 768 / 8 clipTransEncoder text heads on the DH = 128 kernel) are zero-padded in registers -- Q columns >= dh are
                       // loaded as 0, K/V chunks past dh are fetched from a valid address and never influence a stored value, O columns
                       // >= dh are not stored
    // bf16-plane operands of Q K^T (attn_qkp_kernel): NP planes each, [plane][rows][ld] bf16
    const __bf16* Qp; const __bf16* Kp;
    size_t q_plane, k_plane;
    int ldqp, ldkp;
    const __bf16* Vp; int ldvp;        // bf16 V rows (attn_qkp_kernel<DH, 1, true>: P.V on the bf16 matrix cores); nullptr = fp32 V
    size_t v_plane;                    // attn_qkp_kernel<DH, 2, true, true>: Q, K and V as the two fp16 planes of the fp32-split mode (Vp: plane stride v_plane)
    size_t o_plane;                    // out_bf16 == 2: element stride between the two fp16 planes of O (= rows of O x ldo)
    // RAGGED batch (the RAG instantiations; kernels.h mmdm_rag_seq): sequence s owns rows [seq_off[s], seq_off[s] + seq_len[s]) of every operand, lengths
    // differ per sequence and live in device memory (one captured graph serves every batch of the same row bucket).  Tq / Tk are then the
    // LONGEST sequence (grid size only); a query tile past its sequence's end leaves at once.
    const int* seq_off; const int* seq_len;
    const int* seq_order; int order_items;      // kernels.h mmdm_rag_seq::order / items (nullptr: sequences in index order)
};

// Which (sequence, head) pair a workgroup works on.  Uniform layout: XCD x owns the contiguous pairs [x ppx, (x + 1) ppx) -- all sequences cost the same.
// Ragged: pair slots are dealt ROUND-ROBIN over the XCDs (slot j -> XCD j % 8; all query tiles of a pair still share one XCD's L2) and walk the
// sequences LONGEST FIRST through the batch's length order: a launch whose workgroups live 30-80 us must not start a 300-frame sequence last, and
// no XCD may get all the long ones.  Re-numbering only: what a workgroup computes for its (sequence, head, query tile) is unchanged.
template <bool RAG, class P>
__device__ __forceinline__ bool pair_of(const P& p, int xcd, int local, int& seq, int& head) {
    const int slot = RAG ? (local / p.qtiles) * 8 + xcd : xcd * p.pairs_per_xcd + local / p.qtiles;
    if (slot >= p.nseq * p.H) return false;
    head = slot % p.H;
    seq = slot / p.H;
    if constexpr (RAG) {
        if (p.seq_order) {
            const int k = p.nseq / p.order_items;
            seq = (seq % k) * p.order_items + p.seq_order[seq / k];
        }
    }
    return true;
}

// (Tq, Tk, first Q / O row, first K / V row) of a workgroup's (sequence, key sequence): kernel arguments in the uniform layout, two loads each in the ragged one
struct SeqGeom { int Tq, Tk; size_t qrow0, krow0; };
template <bool RAG>
__device__ __forceinline__ SeqGeom seq_geom(const AttnArgs& p, int seq, int kvseq) {
    if constexpr (RAG) return SeqGeom{p.seq_len[seq], p.seq_len[kvseq], (size_t)p.seq_off[seq], (size_t)p.seq_off[kvseq]};
    else return SeqGeom{p.Tq, p.Tk, (size_t)seq * p.Tq, (size_t)kvseq * p.Tk};
}

int g_attn_ablate = 0;
unsigned long long* g_attn_stamps = nullptr;

typedef __bf16 bf16x4a __attribute__((ext_vector_type(4)));

// Store one query row of the P.V accumulators (element (j, r) of o[] is column NJ*lq + tile-column(j) of row q0 + 4g + r: see load_v_row),
// scaled by 1/l, as fp32, bf16 or the two fp16 split planes: 16-byte (fp32) / 8-byte (16-bit) accesses.
template <int DH, class P>
__device__ __forceinline__ void store_o_row(const P& p, const f32x4 (&o)[DH / 16], int r, float inv, size_t row_off, int lq) {
    constexpr int NJ = DH / 16;
#pragma unroll
    for (int h = 0; h < NJ / 4; ++h) {
        const int col = NJ * lq + (DH == 128 ? ((lq >= 8 ? 4 : 0) ^ (4 * h)) : 0);
        if (col >= p.dh) continue;
        const f32x4 y = f32x4{o[4 * h][r], o[4 * h + 1][r], o[4 * h + 2][r], o[4 * h + 3][r]} * inv;
        const size_t off = row_off + col;
        if (p.out_bf16 == 2) {
            mmdm_h4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const _Float16 t = mmdm_split_hi(y[e]); oh[e] = t; ol[e] = mmdm_split_lo(y[e], t); }
            _Float16* op = reinterpret_cast<_Float16*>(p.O) + off;
            const size_t plane = p.o_plane;
            *reinterpret_cast<mmdm_h4*>(op) = oh;
            *reinterpret_cast<mmdm_h4*>(op + plane) = ol;
        } else if (p.out_bf16) {
            const bf16x4a b = {(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
            *reinterpret_cast<bf16x4a*>(reinterpret_cast<__bf16*>(p.O) + off) = b;
        } else {
            *reinterpret_cast<f32x4*>(p.O + off) = y;
        }
    }
}


// LDS image of one stage (KC keys): K rows and V rows are DH floats, unpadded (LDS-DMA writes 1 KiB contiguous pieces);
// bank conflicts are avoided by XOR swizzles applied on the per-lane SOURCE address of the DMA and again on the read:
//   K: 16-byte chunk c of key row r is stored at chunk (c ^ (r & 15))      (conflict-free ds_read_b128 fragment reads)
//   V: 16-byte chunk c of key row r is stored at chunk (c ^ (4 * ((r >> 2) & 1)))  (conflict-free ds_read_b32 operand reads)
// All-reduce over the four 16-lane rows of a wave (the four key groups g of one query lq) WITHOUT the LDS: gfx950's v_permlane16_swap /
// v_permlane32_swap exchange rows / halves between two registers (a' = {a.r0, b.r0, a.r2, b.r2}, b' = {a.r1, b.r1, a.r3, b.r3};
// a' = {a.lo, b.lo}, b' = {a.hi, b.hi}: tools/permlane_probe.hip), so with a = b = x one swap + one VALU op is a butterfly step.  The
// __shfl_xor form is a ds_bpermute: an LDS round trip that queues behind the K / V fragment reads and the LDS-DMA writes of four
// co-resident workgroups -- the softmax stretch of a chunk spent most of its 2 800 cycles in eight of them (tools/attn_timeline.py).
// Inline asm: this compiler's two-result builtin returns the first result twice.  The s_nop cover the VALU -> permlane hazards.
// v_max_f32 / v_max3_f32 as such: fmaxf() of values the compiler cannot prove quiet (MFMA results, permlane outputs) is preceded by a
// canonicalising v_max x, x per operand -- six extra VALU instructions per chunk, and in this loop a VALU instruction costs matrix-pipe time
// (LAB_NOTES.md).  Same result for every input but a signalling NaN, which is returned as it is instead of quieted.
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float rows_max(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = vmax(a, b); b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return vmax(a, b);
}
__device__ __forceinline__ float rows_sum(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = a + b; b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// DIAG: the ablation switches and in-kernel stamps (tools/attn_bench.py ABL=, tools/attn_timeline.py) exist in a second instantiation only.  As
// runtime tests inside the chunk loop they split its basic blocks -- the same thing cost the fp32-split GEMM 14 % -- so the
// production instantiation sees them as compile-time zeros.
//
// QT = query tiles of 16 per wave.  Production runs QT = 1 (a workgroup = 64 queries, four workgroups per CU at 112 registers).  QT = 2 --
// 128 queries per workgroup, tile-major so that the last workgroup's live tiles spread over its waves, K fragments and V rows shared by
// the two tiles, 3 workgroups per (sequence, head) instead of 5 = exactly three rounds of 512 slots at T = 300 -- was built and measured
// in round 3: bit-identical results, but 192 registers leave two waves per SIMD and the kernel is 19 % SLOWER (246.6 vs 207.7 us at
// 64 x 8 x 300 x 128 with warm clocks; 138 vs 129 us at dh = 64, where both forms keep four waves per SIMD): what this kernel needs is
// waves per SIMD, not fewer fixed costs per MFMA (LAB_NOTES.md).  The template parameter stays (the code is the same), QT = 2 is not
// instantiated.  A tile whose 16 queries all lie past Tq is never computed (wave-uniform: `NA` live tiles, one chunk loop per count).
// NW = waves per workgroup.  Production runs NW = 4.  NW = 8 (round 3: 128 queries per workgroup at ONE tile per wave, so registers and the
// four waves per SIMD stay; two 8-wave workgroups per CU share each K / V stage among twice as many waves and T = 300 gives exactly three
// rounds of 512 slots) was measured 13 % SLOWER (236 vs 209 us at 64 x 8 x 300 x 128; bit-identical results): four independent
// workgroups per CU de-synchronise, two 8-wave ones put pairs of lock-stepped waves on every SIMD.  Not instantiated (LAB_NOTES.md).
template <int DH, bool DIAG = false, int QT = 1, int NW = 4, bool RAG = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void attn_mfma_kernel(AttnArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)          // the buffer-resource type of the LDS-DMA builtin exists in the device pass only
    const int ablate = DIAG ? p.ablate : 0;
    unsigned long long* const stamps = DIAG ? p.stamps : nullptr;
    constexpr int KC = KCF;
    constexpr int NJ = DH / 16;                 // d groups of 16 (QK^T) == 16-wide output column tiles (PV)
    static_assert(KC == 16, "one 16-key tile per stage");
    constexpr int CPR = DH / 4;                 // 16-byte chunks per row
    constexpr int RPP = 64 / CPR;               // rows per 1-KiB DMA piece
    constexpr int NPIECE = 2 * KC / RPP;        // pieces per stage (K then V)
    constexpr int NI = NPIECE / NW;             // pieces per wave
    static_assert(NPIECE % NW == 0 && NI >= 1, "pieces of a stage divide evenly over the waves");
    constexpr int STAGE = 2 * KC * DH;          // floats per stage
    constexpr int QBT = QW * NW;                // queries per tile row of the workgroup (one tile per wave)
    constexpr int QBW = QBT * QT;               // queries per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [NST stages][K: KC*DH | V: KC*DH]

    // XCD-aware mapping: blocks b and b+8 share an XCD (private L2), so all query tiles of one (sequence, head) are given to
    // blocks of the same residue mod 8 -- its K/V is then fetched into one L2 once instead of into up to `qtiles` L2s.
    const int bid = blockIdx.x;
    const int local = bid >> 3, xcd = bid & 7;
    const int qt = local % p.qtiles;
    int seq, head;
    if (!pair_of<RAG>(p, xcd, local, seq, head)) return;
    const int kvseq = (seq + p.shift) % p.nseq;
    const SeqGeom G = seq_geom<RAG>(p, seq, kvseq);
    if constexpr (RAG) { if (qt * QBW >= G.Tq) return; }      // (wave-uniform, before any barrier: the whole workgroup leaves)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, g = lane >> 4;
    const int q0 = qt * QBW + wave * QW;        // first query of this wave's tile 0; tile t starts at q0 + QBT * t
    // live tiles of this wave: tiles are ordered by query, so tile t live implies tile t - 1 live
    int nact = 0;
#pragma unroll
    for (int t = 0; t < QT; ++t) nact += (q0 + QBT * t < G.Tq) ? 1 : 0;
    nact = __builtin_amdgcn_readfirstlane(nact);
    unsigned long long t_qk = 0, t_sm = 0, t_pv = 0, t_a = 0;
    if (stamps && tid == 0) {
        stamps[8 * (size_t)bid + 0] = __builtin_amdgcn_s_memrealtime();
        stamps[8 * (size_t)bid + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                                        ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) << 32);   // HW_ID, XCC_ID
    }

    f32x4 qf[QT][NJ];                           // Q fragments: loaded behind the first K / V stage's requests (below)
    f32x4 o[QT][NJ];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) o[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // add_zero_attn: one key with logit 0, value 0 already absorbed in the initial state; without it the state starts empty
    // (m = -inf, l = 0: the first chunk always holds a visible key, so alpha = 2^(-inf) = 0 and no NaN can form).
    const bool nozero = (p.flags & MMDM_ATTN_NO_ZERO_KEY) != 0, causal = (p.flags & MMDM_ATTN_CAUSAL) != 0;
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { m_run[t] = nozero ? -INFINITY : 0.f; l_run[t] = nozero ? 0.f : 1.f; }

    const float* Kg = p.K + G.krow0 * p.ldk + head * p.dh;
    const float* Vg = p.V + G.krow0 * p.ldv + head * p.dh;

    // DMA pieces of this wave: piece pq = wave + 4u; pq < NPIECE/2 -> K rows RPP*pq.., else V rows.  Buffer-addressed (gemm_f32.hip has
    // the measurement): one resource per operand over this (sequence, head)'s rows, a per-lane byte offset that is fixed for the whole
    // launch (row inside the chunk, swizzled 16-byte column) and ONE scalar offset per chunk -- the 64-bit per-lane address arithmetic
    // that used to precede every piece is gone from the loop.  Rows past Tk fall outside the resource and arrive as zeros (their
    // scores are masked to -inf below, so the values never matter).
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Kg), 0, (unsigned)(((size_t)(G.Tk - 1) * p.ldk + p.dh) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vg), 0, (unsigned)(((size_t)(G.Tk - 1) * p.ldv + p.dh) * 4), 0x00020000);
    int voff[NI], dsto[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int pq = wave + NW * u;
        const bool isk = pq < NPIECE / 2;
        const int trow = RPP * (isk ? pq : pq - NPIECE / 2) + lane / CPR;
        const int pos = lane % CPR;
        int src_chunk = isk ? (pos ^ (trow & 15)) : pos;                       // V rows are stored unswizzled (load_v_row)
        if (4 * src_chunk >= p.dh) src_chunk = 0;                              // padded head: any valid address (Q is 0 there / column not stored)
        voff[u] = (trow * (isk ? p.ldk : p.ldv) + 4 * src_chunk) * 4;
        dsto[u] = (isk ? 0 : KC * DH) + RPP * (isk ? pq : pq - NPIECE / 2) * DH;
    }
    // piece u of this wave is a K piece or a V piece: wave-uniform (for NPIECE / 2 % NW == 0 the same for every wave)
    auto stage = [&](int c0, int buf) {
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const bool isk = wave + NW * u < NPIECE / 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isk ? rsK : rsV, (lptr_t)(smem + buf * STAGE + dsto[u]), 16, voff[u], c0 * (isk ? p.ldk : p.ldv) * 4, 0, 0);
        }
    };

    int nchunks = (G.Tk + KC - 1) / KC;
    if (causal) {                               // keys past the workgroup's last query are never visible (same count for all 4 waves: barriers)
        const int last_q = min(qt * QBW + QBW - 1, G.Tq - 1);
        nchunks = min(nchunks, last_q / KC + 1);
    }

    // One chunk for the first NA tiles of this wave.  K fragments and V rows are read once and feed every live tile.
    auto chunk = [&](auto na_c, int c0, const float* Ks, const float* Vs) {
        constexpr int NA = decltype(na_c)::value;
        // S^T tiles: st[t][reg] = score(key = c0 + 4g + reg, query = lq of tile t); K fragments double-buffered in registers
        f32x4 st[NA];
#pragma unroll
        for (int t = 0; t < NA; ++t) st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 kf[2];
        kf[0] = *reinterpret_cast<const f32x4*>(&Ks[lq * DH + 4 * (g ^ lq)]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cb = j & 1;
            if (j + 1 < NJ && !(ablate & 8)) kf[cb ^ 1] = *reinterpret_cast<const f32x4*>(&Ks[lq * DH + 4 * ((4 * (j + 1) + g) ^ lq)]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < NA; ++t)
                    st[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[cb][s], qf[t][j][s], st[t], 0, 0, 0);
            // keep the hand-made double buffer: without the fence the scheduler hoists every fragment read of the chunk to its top,
            // which at two tiles per wave costs 24 more live registers than the 256 of two waves per SIMD (17 spills, reloaded -- with
            // vmcnt(0) -- inside the loop)
            if constexpr (QT > 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NA; ++t) MFMA_SETTLE(st[t]);
        if (stamps) { const unsigned long long tt = __builtin_amdgcn_s_memrealtime(); t_qk += tt - t_a; t_a = tt; }
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int q0t = q0 + QBT * t;
            if (c0 + KC > G.Tk || (causal && c0 + KC - 1 > q0t)) {   // keys past Tk (last chunk) or above the diagonal (wave-uniform branch)
                const int kmax = causal ? min(G.Tk - 1, q0t + lq) : G.Tk - 1;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + 4 * g + r > kmax) st[t][r] = -INFINITY;
            }
            if (!(ablate & 4)) {
                float cmax = vmax3(st[t][0], st[t][1], vmax(st[t][2], st[t][3]));
                cmax = rows_max(cmax);
                // Deferred reference: the running maximum only moves when the chunk's maximum exceeds it by more than 2^8 (scores are in the log2
                // domain), so after the first chunks alpha is exactly 1 for every query of the wave and the rescale below is skipped; until
                // then probabilities up to 2^8 enter the fp32 sums, which changes nothing but the last bits (softmax does not depend on the
                // reference point).  (A speculative form -- probabilities against the current reference first, the cross-lane maximum only
                // when one exceeds 2^8, row sums reduced after the loop: 14 VALU instructions per chunk instead of 40 -- measured the same
                // 200 us at 128 registers: what is left of the softmax stretch is its dependent chain, not its instruction count.)
                const float m_new = cmax > m_run[t] + 8.0f ? cmax : m_run[t];
                const float alpha = EXP2(m_run[t] - m_new);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[t][r] = EXP2(st[t][r] - m_new);
                float lsum = ((st[t][0] + st[t][1]) + st[t][2]) + st[t][3];      // (0 + p0 is p0: the same bits without the add)
                lsum = rows_sum(lsum);
                l_run[t] = l_run[t] * alpha + lsum;
                m_run[t] = m_new;
                // rescale O: accumulator rows are queries 4g + r, whose alpha lives in lanes with (lane&15) == 4g + r.  Once the running maxima
                // of a tile's 16 queries have stopped moving -- the usual case after the first chunks -- every alpha is exactly 1 and the 4
                // cross-lane reads + 4*NJ multiplies are skipped (wave-uniform branch; x * 1.0f is exact, so the result is unchanged).
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                    float ar[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[t][j][r] *= ar[r];   // (the PV MFMAs of the previous chunk retired long ago: barrier + QK^T in between)
                }
            }
        }
        // O[q][n] += sum_key P[q][key] V[key][n]:  A = P (lane-local: st[t][r] is P[q = lq of tile t][key = 4g + r]),
        // B = V[key = 4g + r][this lane's NJ contiguous columns] (load_v_row), shared by the tiles
        if (stamps) { const unsigned long long tt = __builtin_amdgcn_s_memrealtime(); t_sm += tt - t_a; t_a = tt; }
        float vb[2][NJ];
        load_v_row<DH>(&Vs[(4 * g) * DH], lq, vb[0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cb = r & 1;
            if (r + 1 < 4 && !(ablate & 16)) load_v_row<DH>(&Vs[(4 * g + r + 1) * DH], lq, vb[cb ^ 1]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < NA; ++t)
                    o[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[t][r], vb[cb][j], o[t][j], 0, 0, 0);
            if constexpr (QT > 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (stamps) {
#pragma unroll
            for (int t = 0; t < NA; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) MFMA_SETTLE(o[t][j]);
            t_pv += __builtin_amdgcn_s_memrealtime() - t_a;
        }
    };

    // NST-stage ring: chunks 0 .. NST-2 are requested up front; in iteration ci the wave waits until at most the NST-2 newest of its
    // chunks are still in flight (= chunk ci has landed), the raw barrier makes that true for every wave and proves that chunk ci-1 --
    // whose buffer is restaged next -- has been read by all of them (their ds_reads were retired by lgkmcnt(0) before the barrier).
    // __syncthreads() would drain vmcnt(0) here and serialise the ring; the Q-fragment loads above are older than every DMA piece.
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nchunks) stage(t * KC, t);
    static_assert(NST == 2, "the Q loads are younger than the first stage's requests: the loop's first wait must be vmcnt(0)");
    // (the first chunk's K / V rows are on their way: the Q loads below share that latency instead of preceding it -- the prologue was 9 us
    // of a 69 us workgroup at dh = 128, tools/attn_timeline.py)
    // Q fragments (B operand of S^T = K Q^T): lane (q = lq, g) holds Q[q][16j + 4g + s], pre-scaled into the log2 domain.
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int qrow = q0 + QBT * t + lq;
        if (qrow >= G.Tq) qrow = G.Tq - 1;
        const float* qp = p.Q + (G.qrow0 + qrow) * p.ldq + head * p.dh + 4 * g;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool in = 16 * j + 4 * g < p.dh;                       // dh % 4 == 0: a 16-byte group is wholly inside or outside
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + (in ? 16 * j : 0));
            qf[t][j] = in ? v * p.scale2 : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // The chunk loop exists once per number of live tiles (wave-uniform choice OUTSIDE the loop): with both bodies inside one loop the
    // register allocator keeps the state of both alive across the branch (263 registers instead of 184 at DH = 128, QT = 2).
    auto run = [&](auto na_c) {
        constexpr int NA = decltype(na_c)::value;
        // the stage index is a compile-time constant (the loop body exists NST times): the K / V fragment reads then address LDS with their
        // per-lane offset + an immediate instead of one VALU add per read (a VALU instruction in this loop costs matrix-pipe time)
        for (int cb = 0; cb < nchunks; cb += NST) {
#pragma unroll
          for (int cur = 0; cur < NST; ++cur) {
            const int ci = cb + cur;
            if (ci >= nchunks) break;
            const int stg = (cur + NST - 1) % NST;
            const int c0 = ci * KC;
            const int left = nchunks - 1 - ci;                     // chunks requested behind this one (capped at NST - 2 by the ring)
            if (left >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NI) : "memory");
            else if (left == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
            else if (left == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(ablate & 2)) __builtin_amdgcn_s_barrier();
            if (ci + NST - 1 < nchunks && !(ablate & 1)) stage(c0 + (NST - 1) * KC, stg);
            const float* Ks = smem + cur * STAGE;
            const float* Vs = Ks + KC * DH;
            if (stamps) { if (ci == 0 && tid == 0) stamps[8 * (size_t)bid + 1] = __builtin_amdgcn_s_memrealtime(); t_a = __builtin_amdgcn_s_memrealtime(); }
            // a wave without a live tile (T = 300, QT = 2: wave 3 of the last workgroup) stages its pieces and keeps the barriers, nothing else
            if constexpr (NA > 0) chunk(na_c, c0, Ks, Vs);
          }
        }
    };
    if (nact == QT) run(std::integral_constant<int, QT>{});
    else if (QT > 1 && nact == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});

#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) MFMA_SETTLE(o[t][j]);
    if (stamps && tid == 0) {
        stamps[8 * (size_t)bid + 2] = __builtin_amdgcn_s_memrealtime();
        stamps[8 * (size_t)bid + 5] = t_qk; stamps[8 * (size_t)bid + 6] = t_sm; stamps[8 * (size_t)bid + 7] = t_pv;
    }
    // normalise and store: accumulator element (j, r) of tile t belongs to query q0 + QBT t + 4g + r, columns as laid out by load_v_row
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float lr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run[t], 4 * g + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qrow = q0 + QBT * t + 4 * g + r;
            if (qrow >= G.Tq) continue;
            store_o_row<DH>(p, o[t], r, 1.0f / lr[r], (G.qrow0 + qrow) * p.ldo + head * p.dh, lq);
        }
    }
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) stamps[8 * (size_t)bid + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void store_out(const AttnArgs& p, size_t idx, float y) {
    if (p.out_bf16 == 2) {
        _Float16* op = reinterpret_cast<_Float16*>(p.O) + idx;
        const size_t plane = p.o_plane;
        const _Float16 t = mmdm_split_hi(y);
        op[0] = t;
        op[plane] = mmdm_split_lo(y, t);
    } else if (p.out_bf16) {
        reinterpret_cast<__bf16*>(p.O)[idx] = (__bf16)y;
    } else {
        p.O[idx] = y;
    }
}

// Q K^T on the bf16 matrix cores, everything else as attn_mfma_kernel.
// NP = 3: Q and K arrive as exact 3-way bf16 splits of the fp32 projections (written by the QKV GEMM's epilogue) and every score is the
// six-term sum q1k1 + q1k2 + q2k1 + q1k3 + q3k1 + q2k2 accumulated in fp32 -- fp32-accurate scores (gemm_split.hip has the argument) from
// 24 v_mfma_f32_16x16x32_bf16 (16 cycles each) instead of 32 v_mfma_f32_16x16x4_f32 (32 cycles each) per 16 keys x 16 queries x dh = 128.
// NP = 1: plain bf16 Q and K (the bf16 path).  The 16x16 C/D register map is the same for every 16x16 MFMA (col = lane & 15 = query,
// row = 4 * (lane >> 4) + reg = key), so the softmax and the fp32 P.V half of the kernel are untouched; V stays fp32.
// K planes in LDS: [plane][16 keys][DH bf16], 16-byte chunk c of key row r stored at c ^ (r & (chunks_per_row - 1)).
// PVB (NP = 1 only): P.V on the bf16 matrix cores as well.  V arrives as bf16 rows (the projection GEMM's bf16 copy) and is staged
// row-major [16 keys][DH bf16]; the B operand of v_mfma_f32_16x16x16_bf16 -- lane (n, kg) holds V[keys 4kg .. 4kg+3][column n] -- is a
// COLUMN of four rows, which gfx950's ds_read_b64_tr_b16 delivers straight from the row-major image: per 16-lane group it reads a
// 4-row x 16-column block, lane 4q + p supplying the address of row q / columns 4p .. 4p+3 and lane i receiving column i of the four
// rows.  The A operand is this lane's own four probabilities (keys 4g .. 4g+3 of query lq) rounded to bf16; the accumulator map is the
// 16x16 one (column = lane & 15, rows 4 (lane >> 4) + reg), so the running rescale is unchanged and element (j, r) is
// O[query 4g + r][column 16 j + lq].  One MFMA of 16 cycles per 16 output columns and 16 keys instead of four of 32.
// V image swizzle (16-byte chunk c of key row r at c ^ x(r)): 256-byte rows (DH = 128) x = ((r & 3) << 2) | ((r >> 2) & 3), 128-byte rows
// (DH = 64) x = ((r >> 1) & 3) << 1 -- without it the eight rows a 32-lane half reads sit on the same banks.
// H2 (NP = 2, PVB): the fp32-split mode's attention.  Q, K and V arrive as the two fp16 planes the projection GEMM wrote INSTEAD of fp32 rows (same
// bytes), x ~= h + l / 2048 (kernels.h).  Scores: hi += kh*qh, lo += kl*qh + kh*ql (three v_mfma_f32_16x16x32_f16 per 32-deep step), S = hi + lo / 2048;
// softmax in fp32 as everywhere; P.V the same way with the probabilities split on the fly -- and ONE accumulator set: the kernel lives on the number
// of resident waves (LAB_NOTES, round 4), and a second set would cost 32 registers.  With the deferred running maximum at +4 instead of +8 every
// probability is <= 16, so ph * 2048 (<= 32768) is an fp16 number and all three products can be formed AT THE SCALE 2^11:
//   o += pl' * vh + ph * vl' + (ph * 2048) * vh        (pl' = (p - ph) * 2048 and vl' = (v - vh) * 2048 are the lo planes as stored)
// (three v_mfma_f32_16x16x16_f16 per 16 columns), O = o / 2048 / l -- exact scalings.  fp32-accurate like the split GEMMs (the same argument:
// representation error 2^-22 per operand, below the fp32 accumulation error of the 32x32x2 / 16x16x4 fp32 MFMA chains it replaces), at 36 short
// MFMAs per 16-key chunk instead of 64 long ones.
#ifndef QKP_TWO_CHAINS
#define QKP_TWO_CHAINS 1              // (0: the four-chain form of rounds 2-5, for A/B builds: 55.7 vs 52.0 us at 64 x 8 x 300 x 128, 61.8 vs 60.8 at dh = 64)
#endif
template <int DH, int NP, bool PVB = false, bool H2 = false, bool RAG = false>
__global__ __launch_bounds__(256, H2 ? 4 : 2) void attn_qkp_kernel(AttnArgs p) {      // H2: four waves per SIMD (128 registers)
#if defined(__HIP_DEVICE_COMPILE__)          // the buffer-resource type of the LDS-DMA builtin exists in the device pass only
    static_assert(!PVB || NP == 1 || H2, "bf16 P.V goes with bf16 scores");
    static_assert(!H2 || (NP == 2 && PVB), "the fp16 two-plane form covers Q K^T and P.V together");
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    constexpr int NVP = H2 ? 2 : 1;             // V planes (16-bit V)
    constexpr int VPLANE = KC * DH / 2;         // floats per 16-bit V plane
    constexpr int NJ = DH / 16;                 // 16-wide output column tiles (PV)
    constexpr int NS = DH / 32;                 // 32-deep reduction steps of Q K^T
    constexpr int CPRK = DH / 8;                // 16-byte chunks per K row (bf16)
    constexpr int RPPK = 64 / CPRK;             // K rows per 1-KiB DMA piece
    constexpr int NPK = KC / RPPK;              // pieces per K plane
    constexpr int CPR = PVB ? DH / 8 : DH / 4;  // 16-byte chunks per V row (bf16 / fp32)
    constexpr int RPP = 64 / CPR;
    constexpr int NPV = KC / RPP;
    constexpr int NPIECE = NP * NPK + NVP * NPV;
    constexpr int NI = (NPIECE + 3) / 4;
    constexpr int KPLANE = KC * DH / 2;         // floats per K plane
    constexpr int STAGE = NP * KPLANE + (PVB ? NVP * VPLANE : KC * DH);
    constexpr int NT = NP == 3 ? 6 : 1;
    constexpr int TK[6] = {1, 2, 0, 1, 0, 0}, TQ[6] = {1, 0, 2, 0, 1, 0};      // (K plane, Q plane) per term, small terms first
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2 stages][K planes | V]

    const int bid = blockIdx.x;
    const int local = bid >> 3, xcd = bid & 7;
    const int qt = local % p.qtiles;
    int seq, head;
    if (!pair_of<RAG>(p, xcd, local, seq, head)) return;
    const int kvseq = (seq + p.shift) % p.nseq;
    const SeqGeom G = seq_geom<RAG>(p, seq, kvseq);
    if constexpr (RAG) { if (qt * QB >= G.Tq) return; }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, g = lane >> 4;
    const int q0 = qt * QB + wave * QW;

    // Q fragments (B operand): lane (q = lq, g) holds Q[q][32 s + 8 g .. + 7] of every plane
    bf16x8 qf[NP][NS];
    {
        int qrow = q0 + lq;
        if (qrow >= G.Tq) qrow = G.Tq - 1;
        const __bf16* qp = p.Qp + (G.qrow0 + qrow) * p.ldqp + head * DH + 8 * g;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int s = 0; s < NS; ++s) qf[pl][s] = *reinterpret_cast<const bf16x8*>(qp + (size_t)pl * p.q_plane + 32 * s);
    }

    f32x4 o[NJ];                                // (H2: at the scale 2^11)
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool nozero = (p.flags & MMDM_ATTN_NO_ZERO_KEY) != 0, causal = (p.flags & MMDM_ATTN_CAUSAL) != 0;
    float m_run = nozero ? -INFINITY : 0.f, l_run = nozero ? 0.f : 1.f;

    const __bf16* Kg = p.Kp + G.krow0 * p.ldkp + head * DH;
    const float* Vg = p.V + G.krow0 * p.ldv + head * DH;

    // DMA pieces of this wave (piece pq = wave + 4u): buffer-addressed as in attn_mfma_kernel -- one resource per piece over this (sequence,
    // head)'s rows of its operand (a K plane, or V), a per-lane byte offset that is fixed for the whole launch and ONE scalar offset per
    // chunk.  (The pointer form computed a clamped row, a 64-bit product and a 64-bit sum per lane, piece and chunk: 40 VALU instructions
    // per chunk beside 12 short MFMAs.)  Rows past Tk fall outside the resource and arrive as zeros; their scores are masked below.
    __amdgpu_buffer_rsrc_t rsp[NI];
    int voff[NI], dsto[NI], rstep[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int pq = wave + 4 * u;                                   // wave-uniform
        const bool isk = pq < NP * NPK;
        const int pl = isk ? pq / NPK : (pq - NP * NPK) / NPV, pp = isk ? pq % NPK : (pq - NP * NPK) % NPV;       // (K or V) plane, piece inside it
        const int trow = isk ? RPPK * pp + lane / CPRK : RPP * pp + lane / CPR;
        const int pos = isk ? lane % CPRK : lane % CPR;
        const void* base; unsigned bytes;
        if (isk) {
            base = Kg + (size_t)pl * p.k_plane; bytes = (unsigned)(((size_t)(G.Tk - 1) * p.ldkp + DH) * 2);
            voff[u] = (trow * p.ldkp + 8 * (pos ^ (trow & (CPRK - 1)))) * 2; rstep[u] = p.ldkp * 2;
            dsto[u] = pl * KPLANE + RPPK * pp * (DH / 2);
        } else if constexpr (PVB) {
            const int xv = DH == 128 ? (((trow & 3) << 2) | ((trow >> 2) & 3)) : (((trow >> 1) & 3) << 1);
            base = p.Vp + (H2 ? (size_t)pl * p.v_plane : 0) + G.krow0 * p.ldvp + head * DH; bytes = (unsigned)(((size_t)(G.Tk - 1) * p.ldvp + DH) * 2);
            voff[u] = (trow * p.ldvp + 8 * (pos ^ xv)) * 2; rstep[u] = p.ldvp * 2;
            dsto[u] = NP * KPLANE + (H2 ? pl * VPLANE : 0) + RPP * pp * (DH / 2);
        } else {
            base = Vg; bytes = (unsigned)(((size_t)(G.Tk - 1) * p.ldv + DH) * 4);
            voff[u] = (trow * p.ldv + 4 * pos) * 4; rstep[u] = p.ldv * 4;              // V rows are stored unswizzled (load_v_row)
            dsto[u] = NP * KPLANE + RPP * pp * DH;
        }
        rsp[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
    }
    auto stage = [&](int c0, int buf) {
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            if (wave + 4 * u >= NPIECE) break;                         // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp[u], (lptr_t)(smem + buf * STAGE + dsto[u]), 16, voff[u], c0 * rstep[u], 0, 0);
        }
    };

    int nchunks = (G.Tk + KC - 1) / KC;
    if (causal) {
        const int last_q = min(qt * QB + QB - 1, G.Tq - 1);
        nchunks = min(nchunks, last_q / KC + 1);
    }
    stage(0, 0);
    // the stage index is a compile-time constant (the body exists twice): fragment reads address LDS with an immediate
    for (int cb2 = 0; cb2 < nchunks; cb2 += 2) {
#pragma unroll
      for (int cur = 0; cur < 2; ++cur) {
        const int ci = cb2 + cur;
        if (ci >= nchunks) break;
        const int c0 = ci * KC;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ci + 1 < nchunks) stage(c0 + KC, cur ^ 1);
        const float* Ks = smem + cur * STAGE;
        const float* Vs = Ks + NP * KPLANE;

        if (q0 >= G.Tq) continue;        // a wave with no query inside Tq keeps staging and the barriers, nothing else (attn_mfma_kernel)
        // S^T tile (16 keys x 16 queries): one accumulator per 32-deep reduction step so that consecutive MFMAs are independent
        f32x4 sa[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) sa[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[NP][NS];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                kf[pl][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Ks[pl * KPLANE + lq * (DH / 2) + 4 * ((4 * s + g) ^ (lq & (CPRK - 1)))]));
        f32x4 st[1];
        if constexpr (H2) {
            // two accumulator chains per sum (even / odd 32-deep steps): 16 registers instead of 32, dependent MFMAs two apart
            f32x4 sl[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            auto hb = [](const bf16x8& v) { return __builtin_bit_cast(h16x8, v); };
#pragma unroll
            for (int s = 0; s < NS; ++s) sl[s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb(kf[1][s]), hb(qf[0][s]), sl[s & 1], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) sl[s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb(kf[0][s]), hb(qf[1][s]), sl[s & 1], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) sa[s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb(kf[0][s]), hb(qf[0][s]), sa[s & 1], 0, 0, 0);
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sa[0]), "+v"(sa[1]), "+v"(sl[0]), "+v"(sl[1]));      // MFMA_SETTLE for the four chains at once
            const f32x4 lo = sl[0] + sl[1];
            st[0] = sa[0] + sa[1];
#pragma unroll
            for (int r = 0; r < 4; ++r) st[0][r] = __builtin_fmaf(lo[r], MMDM_SPLIT_INV, st[0][r]);
        } else if constexpr (NP == 1 && QKP_TWO_CHAINS) {
            // one plane (the all-bf16 form of configs[4]): two accumulator chains (even / odd 32-deep steps: dependent MFMAs two apart, as in the H2 form)
            // and ONE settle for both -- NS separate settles are 24 wait states each, 96 idle cycles of a chunk whose MFMAs take 128
#pragma unroll
            for (int s = 0; s < NS; ++s) sa[s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][s], qf[0][s], sa[s & 1], 0, 0, 0);
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sa[0]), "+v"(sa[1]));
            st[0] = sa[0] + sa[1];
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                sa[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[NP == 3 ? TK[t] : 0][s], qf[NP == 3 ? TQ[t] : 0][s], sa[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) MFMA_SETTLE(sa[s]);
        st[0] = sa[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) st[0] += sa[s];
        }
        st[0] *= p.scale2;

        if (c0 + KC > G.Tk || (causal && c0 + KC - 1 > q0)) {
            const int kmax = causal ? min(G.Tk - 1, q0 + lq) : G.Tk - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (c0 + 4 * g + r > kmax) st[0][r] = -INFINITY;
        }
        float cmax = vmax3(st[0][0], st[0][1], vmax(st[0][2], st[0][3]));
        cmax = rows_max(cmax);
        const float m_new = cmax > m_run + (H2 ? 4.0f : 8.0f) ? cmax : m_run;        // deferred reference, as in attn_mfma_kernel (H2: p <= 2^4, see above)
        const float alpha = EXP2(m_run - m_new);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[0][r] = EXP2(st[0][r] - m_new);
        float lsum = ((st[0][0] + st[0][1]) + st[0][2]) + st[0][3];
        lsum = rows_sum(lsum);
        l_run = l_run * alpha + lsum;
        m_run = m_new;

        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {          // exact skip: x * 1.0f == x
            float ar[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[j][r] *= ar[r];
        }

        if constexpr (H2) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
            mmdm_h4 ph, pl, phs;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const _Float16 t = mmdm_split_hi(st[0][r]);
                ph[r] = t; pl[r] = mmdm_split_lo(st[0][r], t); phs[r] = (_Float16)((float)t * MMDM_SPLIT_SCALE);       // exact: p <= 16
            }
            const int vrow = 4 * g + (lq >> 2), pp = lq & 3;
            const int xv = DH == 128 ? (((vrow & 3) << 2) | ((vrow >> 2) & 3)) : (((vrow >> 1) & 3) << 1);
            const char* vbase = reinterpret_cast<const char*>(Vs) + vrow * (DH * 2) + 8 * (pp & 1);
            constexpr int JG = 4;                  // column tiles per group: the three MFMAs on an accumulator are JG apart, 4 JG registers of V fragments live
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += JG) {
                mmdm_h4 vh[JG], vl[JG];
#pragma unroll
                for (int j = 0; j < JG; ++j) {
                    vh[j] = __builtin_bit_cast(mmdm_h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vbase + 16 * ((2 * (j0 + j) + (pp >> 1)) ^ xv))));
                    vl[j] = __builtin_bit_cast(mmdm_h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vbase + VPLANE * 4 + 16 * ((2 * (j0 + j) + (pp >> 1)) ^ xv))));
                }
#pragma unroll
                for (int j = 0; j < JG; ++j) o[j0 + j] = __builtin_amdgcn_mfma_f32_16x16x16f16(pl, vh[j], o[j0 + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < JG; ++j) o[j0 + j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ph, vl[j], o[j0 + j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < JG; ++j) o[j0 + j] = __builtin_amdgcn_mfma_f32_16x16x16f16(phs, vh[j], o[j0 + j], 0, 0, 0);
            }
        } else if constexpr (PVB) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
            const bf16x4a pb = {(__bf16)st[0][0], (__bf16)st[0][1], (__bf16)st[0][2], (__bf16)st[0][3]};
            const s16x4 pa = __builtin_bit_cast(s16x4, pb);
            // transposed-read address of this lane: row 4g + q of the stage, columns 16 j + 4 pp .. + 3 (q = (lane & 15) >> 2, pp = lane & 3)
            const int vrow = 4 * g + (lq >> 2), pp = lq & 3;
            const int xv = DH == 128 ? (((vrow & 3) << 2) | ((vrow >> 2) & 3)) : (((vrow >> 1) & 3) << 1);
            const char* vbase = reinterpret_cast<const char*>(Vs) + vrow * (DH * 2) + 8 * (pp & 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vbase + 16 * ((2 * j + (pp >> 1)) ^ xv)));
                o[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, vb, o[j], 0, 0, 0);
            }
        } else {
        float vb[2][NJ];
        load_v_row<DH>(&Vs[(4 * g) * DH], lq, vb[0]);
#pragma unroll
        for (int idx = 0; idx < 4; ++idx) {
            const int r = idx & 3, cb = idx & 1;
            if (idx + 1 < 4) load_v_row<DH>(&Vs[(4 * g + idx + 1) * DH], lq, vb[cb ^ 1]);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[0][r], vb[cb][j], o[j], 0, 0, 0);
        }
        }
      }
    }

#pragma unroll
    for (int j = 0; j < NJ; ++j) MFMA_SETTLE(o[j]);
    float lr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run, 4 * g + r);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * g + r;
        if (qrow >= G.Tq) continue;
        if constexpr (PVB) {                        // element (j, r) is column 16 j + lq of row 4g + r
            const float inv = (1.0f / lr[r]) * (H2 ? MMDM_SPLIT_INV : 1.0f);       // H2: the accumulators carry the scale 2^11 (an exact scaling)
            const size_t off = (G.qrow0 + qrow) * p.ldo + head * DH + lq;
#pragma unroll
            for (int j = 0; j < NJ; ++j) store_out(p, off + 16 * j, o[j][r] * inv);
        } else {
            store_o_row<DH>(p, o, r, 1.0f / lr[r], (G.qrow0 + qrow) * p.ldo + head * DH, lq);
        }
    }
#endif
}

// ---- the all-bf16 attention on 32-KEY chunks (round 6; configs[4]) -------------------------------------------------------------------------------------
// attn_qkp_kernel<DH, 1, true> spends a 16-key chunk on 128 matrix-pipe cycles (4 Q K^T + 8 P.V short MFMAs) beside ~ 80 other instructions, one
// s_waitcnt vmcnt(0) + barrier and two cross-row reductions: the matrix pipe is 23-29 % busy.  Here a chunk is 32 keys -- two 16 x 16 score tiles side by
// side, ONE running-maximum / rescale / row-sum sequence and one barrier for both, and P.V on v_mfma_f32_16x16x32_bf16: the lane's eight probabilities
// (keys 4g .. 4g+3 of tile 0 | of tile 1) against the two transposed V reads of the same keys -- the k index of the 32-deep MFMA is permuted identically on
// both operands, so every product pairs the same key.  Same MFMA cycles per key, half the fixed work.  A last chunk whose second tile lies past Tk skips
// that tile's score MFMAs (its probabilities are exact zeros).  Staging, swizzles, query ownership and output map as attn_qkp_kernel; results differ from
// the 16-key form in the last bits (other rescale points, other summation order) -- the bf16 mode's stated accuracy, not a parity path.
#ifndef B32_WAVES
#define B32_WAVES 2
#endif
template <int DH, bool RAG = false>
__global__ __launch_bounds__(256, B32_WAVES) void attn_b32_kernel(AttnArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    constexpr int KC2 = 32;
    constexpr int NJ = DH / 16, NS = DH / 32;
    constexpr int CPRK = DH / 8, RPPK = 64 / CPRK, NPK = KC2 / RPPK;       // 16-byte chunks per bf16 row, rows per 1-KiB DMA piece, pieces per operand
    constexpr int NPIECE = 2 * NPK, NI = NPIECE / 4;
    constexpr int PLANE = KC2 * DH / 2;                                    // floats per operand image (K or V)
    constexpr int STAGE = 2 * PLANE;
    static_assert(NPIECE % 4 == 0, "pieces divide over the four waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];          // [2 stages][K | V]

    const int bid = blockIdx.x;
    const int local = bid >> 3, xcd = bid & 7;
    const int qt = local % p.qtiles;
    int seq, head;
    if (!pair_of<RAG>(p, xcd, local, seq, head)) return;
    const int kvseq = (seq + p.shift) % p.nseq;
    const SeqGeom G = seq_geom<RAG>(p, seq, kvseq);
    if constexpr (RAG) { if (qt * QB >= G.Tq) return; }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, g = lane >> 4;
    const int q0 = qt * QB + wave * QW;

    bf16x8 qf[NS];
    {
        int qrow = q0 + lq;
        if (qrow >= G.Tq) qrow = G.Tq - 1;
        const __bf16* qp = p.Qp + (G.qrow0 + qrow) * p.ldqp + head * DH + 8 * g;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 32 * s);
    }
    f32x4 o[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool nozero = (p.flags & MMDM_ATTN_NO_ZERO_KEY) != 0, causal = (p.flags & MMDM_ATTN_CAUSAL) != 0;
    float m_run = nozero ? -INFINITY : 0.f, l_run = nozero ? 0.f : 1.f;

    __amdgpu_buffer_rsrc_t rsp[NI];
    int voff[NI], dsto[NI], rstep[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int pq = wave + 4 * u;                                   // wave-uniform
        const bool isk = pq < NPK;
        const int pp = isk ? pq : pq - NPK;
        const int trow = RPPK * pp + lane / CPRK, pos = lane % CPRK;
        const void* base; unsigned bytes;
        if (isk) {
            base = p.Kp + G.krow0 * p.ldkp + head * DH; bytes = (unsigned)(((size_t)(G.Tk - 1) * p.ldkp + DH) * 2);
            voff[u] = (trow * p.ldkp + 8 * (pos ^ (trow & (CPRK - 1)))) * 2; rstep[u] = p.ldkp * 2;
            dsto[u] = RPPK * pp * (DH / 2);
        } else {
            const int xv = DH == 128 ? (((trow & 3) << 2) | ((trow >> 2) & 3)) : (((trow >> 1) & 3) << 1);
            base = p.Vp + G.krow0 * p.ldvp + head * DH; bytes = (unsigned)(((size_t)(G.Tk - 1) * p.ldvp + DH) * 2);
            voff[u] = (trow * p.ldvp + 8 * (pos ^ xv)) * 2; rstep[u] = p.ldvp * 2;
            dsto[u] = PLANE + RPPK * pp * (DH / 2);
        }
        rsp[u] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
    }
    auto stage = [&](int c0, int buf) {
#pragma unroll
        for (int u = 0; u < NI; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp[u], (lptr_t)(smem + buf * STAGE + dsto[u]), 16, voff[u], c0 * rstep[u], 0, 0);
    };

    int nchunks = (G.Tk + KC2 - 1) / KC2;
    if (causal) {
        const int last_q = min(qt * QB + QB - 1, G.Tq - 1);
        nchunks = min(nchunks, last_q / KC2 + 1);
    }
    // transposed-read address of this lane inside a 16-key tile: row 4g + (lq >> 2), columns 16 j + 4 (lq & 3) .. + 3 (attn_qkp_kernel)
    const int vrow = 4 * g + (lq >> 2), pp = lq & 3;
    const int xv = DH == 128 ? (((vrow & 3) << 2) | ((vrow >> 2) & 3)) : (((vrow >> 1) & 3) << 1);      // (the same for row 16 + vrow)
    stage(0, 0);
    for (int cb2 = 0; cb2 < nchunks; cb2 += 2) {
#pragma unroll
      for (int cur = 0; cur < 2; ++cur) {
        const int ci = cb2 + cur;
        if (ci >= nchunks) break;
        const int c0 = ci * KC2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ci + 1 < nchunks) stage(c0 + KC2, cur ^ 1);
        const float* Ks = smem + cur * STAGE;
        const char* Vs = reinterpret_cast<const char*>(Ks + PLANE);
        if (q0 >= G.Tq) continue;                  // a wave with no query inside Tq keeps staging and the barriers, nothing else
        const bool two = c0 + 16 < G.Tk;           // (wave-uniform) the second score tile holds at least one key
        f32x4 sa[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
        bf16x8 kf[2][NS];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                kf[t][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(&Ks[(16 * t + lq) * (DH / 2) + 4 * ((4 * s + g) ^ (lq & (CPRK - 1)))]));
#pragma unroll
        for (int s = 0; s < NS; ++s) sa[0][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][s], qf[s], sa[0][s & 1], 0, 0, 0);
        if (two) {
#pragma unroll
            for (int s = 0; s < NS; ++s) sa[1][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][s], qf[s], sa[1][s & 1], 0, 0, 0);
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(sa[0][0]), "+v"(sa[0][1]), "+v"(sa[1][0]), "+v"(sa[1][1]));
        f32x4 st[2];
        st[0] = (sa[0][0] + sa[0][1]) * p.scale2;
        st[1] = (sa[1][0] + sa[1][1]) * p.scale2;
        if (c0 + KC2 > G.Tk || (causal && c0 + KC2 - 1 > q0)) {
            const int kmax = causal ? min(G.Tk - 1, q0 + lq) : G.Tk - 1;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + 16 * t + 4 * g + r > kmax) st[t][r] = -INFINITY;
        }
        float cmax = vmax(vmax3(st[0][0], st[0][1], st[0][2]), vmax3(st[0][3], st[1][0], vmax3(st[1][1], st[1][2], st[1][3])));
        cmax = rows_max(cmax);
        const float m_new = cmax > m_run + 8.0f ? cmax : m_run;         // deferred reference, as in attn_mfma_kernel
        const float alpha = EXP2(m_run - m_new);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[t][r] = EXP2(st[t][r] - m_new);
        float lsum = (((st[0][0] + st[0][1]) + st[0][2]) + st[0][3]) + (((st[1][0] + st[1][1]) + st[1][2]) + st[1][3]);
        lsum = rows_sum(lsum);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {          // exact skip: x * 1.0f == x
            float ar[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[j][r] *= ar[r];
        }
        const bf16x8 pa = {(__bf16)st[0][0], (__bf16)st[0][1], (__bf16)st[0][2], (__bf16)st[0][3], (__bf16)st[1][0], (__bf16)st[1][1], (__bf16)st[1][2], (__bf16)st[1][3]};
        const char* vb0 = Vs + vrow * (DH * 2) + 8 * (pp & 1);
        const char* vb1 = vb0 + 16 * (DH * 2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int co = 16 * ((2 * j + (pp >> 1)) ^ xv);
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb0 + co));
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb1 + co));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 vb = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, __builtin_bit_cast(bf16x8, vb), o[j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) MFMA_SETTLE(o[j]);
    float lr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run, 4 * g + r);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * g + r;
        if (qrow >= G.Tq) continue;
        const float inv = 1.0f / lr[r];
        const size_t off = (G.qrow0 + qrow) * p.ldo + head * DH + lq;       // element (j, r) is column 16 j + lq of row 4g + r
#pragma unroll
        for (int j = 0; j < NJ; ++j) store_out(p, off + 16 * j, o[j][r] * inv);
    }
#endif
}

int g_attn_kc32 = 1;          // mmdm_diag_set "attn_kc32": 0 = the 16-key form of rounds 2-5 for the all-bf16 attention (A/B, tests)

template <int DH>
int launch_b32(const AttnArgs& a, hipStream_t st) {
    constexpr int smem_bytes = 2 * 2 * (32 * DH / 2) * 4;
    if (a.seq_off) hipLaunchKernelGGL((attn_b32_kernel<DH, true>), dim3(8 * a.pairs_per_xcd * a.qtiles), dim3(256), smem_bytes, st, a);
    else hipLaunchKernelGGL((attn_b32_kernel<DH>), dim3(8 * a.pairs_per_xcd * a.qtiles), dim3(256), smem_bytes, st, a);
    return mmdm_check_launch("attn_b32");
}

template <int DH, int NP, bool PVB = false, bool H2 = false>
constexpr int qkp_smem() { return 2 * (NP * KC * DH / 2 + (PVB ? (H2 ? 2 : 1) * KC * DH / 2 : KC * DH)) * 4; }

template <int DH, int NP, bool PVB = false, bool H2 = false>
int launch_qkp(const AttnArgs& a, hipStream_t st) {
    constexpr int smem_bytes = qkp_smem<DH, NP, PVB, H2>();
    if (a.seq_off) {
        // ragged batches: the production forms only (all-bf16 and the fp16 planes); the three-plane / fp32-V experiments stay uniform
        if constexpr (PVB) hipLaunchKernelGGL((attn_qkp_kernel<DH, NP, PVB, H2, true>), dim3(8 * a.pairs_per_xcd * a.qtiles), dim3(256), smem_bytes, st, a);
        else return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "attention on ragged batches: this operand form has no ragged instantiation");
    } else hipLaunchKernelGGL((attn_qkp_kernel<DH, NP, PVB, H2>), dim3(8 * a.pairs_per_xcd * a.qtiles), dim3(256), smem_bytes, st, a);
    return mmdm_check_launch("attn_qkp");
}

// Small-head fallback (dh in {4,8,16,32}): one thread per (sequence, head, query); used by tiny test configurations.
template <int DH>
__global__ void attn_small_kernel(AttnArgs p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = p.nseq * p.H * p.Tq;
    if (idx >= total) return;
    const int q = idx % p.Tq;
    const int head = (idx / p.Tq) % p.H;
    const int seq = idx / (p.Tq * p.H);
    const int kvseq = (seq + p.shift) % p.nseq;
    float qv[DH], acc[DH];
    const float* qp = p.Q + ((size_t)seq * p.Tq + q) * p.ldq + head * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) { qv[d] = qp[d] * p.scale; acc[d] = 0.f; }
    const bool nozero = (p.flags & MMDM_ATTN_NO_ZERO_KEY) != 0;
    float m = nozero ? -INFINITY : 0.f, l = nozero ? 0.f : 1.f;
    const int kend = (p.flags & MMDM_ATTN_CAUSAL) ? min(p.Tk, q + 1) : p.Tk;
    for (int k = 0; k < kend; ++k) {
        const float* kp = p.K + ((size_t)kvseq * p.Tk + k) * p.ldk + head * DH;
        const float* vp = p.V + ((size_t)kvseq * p.Tk + k) * p.ldv + head * DH;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) s += qv[d] * kp[d];
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), e = expf(s - mn);
        l = l * a + e;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = acc[d] * a + e * vp[d];
        m = mn;
    }
    const size_t off = ((size_t)seq * p.Tq + q) * p.ldo + head * DH;
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < DH; ++d) store_out(p, off + d, acc[d] * inv);
}

// Any-head-size fallback (dh <= 256, e.g. the 96-wide heads of the 768/8 clipTransEncoder text heads): one wavefront per
// (sequence, head, query), lanes over the head dimension, one wave reduction per key.  Only the short text sequences use it.
__global__ __launch_bounds__(256) void attn_wave_kernel(AttnArgs p, int dh) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = p.nseq * p.H * p.Tq;
    if (idx >= total) return;
    const int lane = threadIdx.x & 63;
    const int q = idx % p.Tq;
    const int head = (idx / p.Tq) % p.H;
    const int seq = idx / (p.Tq * p.H);
    const int kvseq = (seq + p.shift) % p.nseq;
    float qv[4], acc[4];
    const float* qp = p.Q + ((size_t)seq * p.Tq + q) * p.ldq + head * dh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        qv[i] = d < dh ? qp[d] * p.scale : 0.f;
        acc[i] = 0.f;
    }
    const bool nozero = (p.flags & MMDM_ATTN_NO_ZERO_KEY) != 0;
    float m = nozero ? -INFINITY : 0.f, l = nozero ? 0.f : 1.f;
    const int kend = (p.flags & MMDM_ATTN_CAUSAL) ? min(p.Tk, q + 1) : p.Tk;
    for (int k = 0; k < kend; ++k) {
        const float* kp = p.K + ((size_t)kvseq * p.Tk + k) * p.ldk + head * dh;
        const float* vp = p.V + ((size_t)kvseq * p.Tk + k) * p.ldv + head * dh;
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = lane + 64 * i;
            if (d < dh) sdot += qv[i] * kp[d];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sdot += __shfl_xor(sdot, o);
        const float mn = fmaxf(m, sdot);
        const float a = expf(m - mn), e = expf(sdot - mn);
        l = l * a + e;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = lane + 64 * i;
            if (d < dh) acc[i] = acc[i] * a + e * vp[d];
        }
        m = mn;
    }
    const size_t off = ((size_t)seq * p.Tq + q) * p.ldo + head * dh;
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d >= dh) continue;
        store_out(p, off + d, acc[i] * inv);
    }
}

template <int DH>
constexpr int attn_smem() { return NST * 2 * KCF * DH * 4; }

template <int DH>
int launch_mfma(const AttnArgs& a, hipStream_t st) {
    const dim3 grid(8 * a.pairs_per_xcd * a.qtiles), block(256);
    if (a.seq_off) hipLaunchKernelGGL((attn_mfma_kernel<DH, false, 1, 4, true>), grid, block, attn_smem<DH>(), st, a);
    else if (a.ablate || a.stamps) hipLaunchKernelGGL((attn_mfma_kernel<DH, true>), grid, block, attn_smem<DH>(), st, a);
    else hipLaunchKernelGGL((attn_mfma_kernel<DH>), grid, block, attn_smem<DH>(), st, a);
    return mmdm_check_launch("attn_mfma");
}

template <int DH>
int launch_small(const AttnArgs& a, hipStream_t st) {
    const int total = a.nseq * a.H * a.Tq;
    hipLaunchKernelGGL((attn_small_kernel<DH>), dim3((total + 127) / 128), dim3(128), 0, st, a);
    return mmdm_check_launch("attn_small");
}

}  // namespace

int mmdm_attn_init(void) {
    hipError_t e = hipSuccess;
    const void* fns[4] = {reinterpret_cast<const void*>(&attn_mfma_kernel<128, false>), reinterpret_cast<const void*>(&attn_mfma_kernel<128, true>),
                          reinterpret_cast<const void*>(&attn_mfma_kernel<64, false>), reinterpret_cast<const void*>(&attn_mfma_kernel<64, true>)};
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, i < 2 ? attn_smem<128>() : attn_smem<64>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_qkp_kernel<128, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, qkp_smem<128, 3>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_qkp_kernel<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, qkp_smem<64, 3>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_qkp_kernel<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, qkp_smem<128, 1>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_qkp_kernel<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, qkp_smem<64, 1>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<128, false, 1, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, attn_smem<128>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<64, false, 1, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, attn_smem<64>());
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(attn): %s", hipGetErrorString(e));
    return MMDM_OK;
}

extern "C" int mmdm_attention_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                  int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    return mmdm_attention_ex(Q, ldq, K, ldk, V, ldv, O, ldo, 0, nseq, Tq, Tk, H, dh, kv_seq_shift, stream);
}

extern "C" int mmdm_attention_ex(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* Ov, int ldo, int out_bf16,
                                 int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    return mmdm_attention_opts(Q, ldq, K, ldk, V, ldv, Ov, ldo, out_bf16, 0, nseq, Tq, Tk, H, dh, kv_seq_shift, stream);
}

extern "C" int mmdm_attention_opts(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* Ov, int ldo, int out_bf16,
                                   int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    return mmdm_attention_opts_rag(Q, ldq, K, ldk, V, ldv, Ov, ldo, out_bf16, flags, nseq, Tq, Tk, H, dh, kv_seq_shift, nullptr, stream);
}

extern "C" int mmdm_attention_ragged_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                         int nseq, const int* seq_off, const int* seq_len, int max_len, int total_rows, int H, int dh, int kv_seq_shift, void* stream) {
    if (!seq_off || !seq_len || max_len <= 0 || total_rows <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_ragged_f32: bad sequence description");
    const mmdm_rag_seq rg{seq_off, seq_len, total_rows, max_len};
    return mmdm_attention_opts_rag(Q, ldq, K, ldk, V, ldv, O, ldo, 0, 0, nseq, max_len, max_len, H, dh, kv_seq_shift, &rg, stream);
}

// rg != nullptr: ragged batch (kernels.h mmdm_rag_seq); Tq = Tk = the longest sequence
int mmdm_attention_opts_rag(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* Ov, int ldo, int out_bf16,
                            int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, const mmdm_rag_seq* rg, void* stream) {
    float* O = static_cast<float*>(Ov);
    if (nseq == 0 || Tq == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    if (!Q || !K || !V || !O || nseq < 0 || Tq < 0 || Tk <= 0 || H <= 0 || dh <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: bad shape nseq=%d Tq=%d Tk=%d H=%d dh=%d", nseq, Tq, Tk, H, dh);
    if (flags & ~(MMDM_ATTN_NO_ZERO_KEY | MMDM_ATTN_CAUSAL)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_opts: unknown flags 0x%x", flags);
    if ((flags & MMDM_ATTN_CAUSAL) && (Tq != Tk || !(flags & MMDM_ATTN_NO_ZERO_KEY)))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_opts: the causal mask needs Tq == Tk and no zero key");
    if (ldq < H * dh || ldk < H * dh || ldv < H * dh || ldo < H * dh)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: row strides must cover H*dh=%d", H * dh);
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.Vp = nullptr; a.ldvp = 0;
    a.nseq = nseq; a.Tq = Tq; a.Tk = Tk; a.H = H; a.out_bf16 = out_bf16; a.flags = flags; a.dh = dh; a.ablate = g_attn_ablate; a.stamps = g_attn_stamps;
    a.Qp = a.Kp = nullptr; a.q_plane = a.k_plane = 0; a.ldqp = a.ldkp = 0;
    a.seq_off = rg ? rg->off : nullptr; a.seq_len = rg ? rg->len : nullptr;
    a.seq_order = (rg && rg->order && rg->items > 0 && nseq % rg->items == 0) ? rg->order : nullptr; a.order_items = rg ? rg->items : 0;
    a.o_plane = (size_t)(rg ? (size_t)rg->total_rows : (size_t)nseq * Tq) * ldo;
    if (rg && (flags || !(dh == 128 || dh == 64))) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "attention on ragged batches: head sizes 64 / 128, zero key, no mask (dh=%d flags=0x%x)", dh, flags);
    a.shift = ((kv_seq_shift % nseq) + nseq) % nseq;
    a.qtiles = (Tq + QB - 1) / QB;
    a.pairs_per_xcd = (nseq * H + 7) / 8;
    a.scale = 1.0f / sqrtf((float)dh);
    a.scale2 = a.scale * 1.4426950408889634f;     // scores kept in the log2 domain: softmax via v_exp_f32 (2^x)
    hipStream_t st = static_cast<hipStream_t>(stream);
    // MFMA kernel: head widths 64 and 128 natively; any other multiple of 4 in (32, 128] zero-padded to the next of the two (AttnArgs::dh)
    if (dh == 128 || dh == 64 || (dh > 32 && dh < 128 && (dh & 3) == 0)) {
        const bool al = ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15) == 0 &&
                        ((ldq | ldk | ldv) & 3) == 0;
        if (!al) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: Q/K/V must be 16-byte aligned with row strides %% 4 == 0");
        if ((reinterpret_cast<uintptr_t>(O) & 15) || (ldo & 3)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: O must be 16-byte aligned with a row stride %% 4 == 0");
        return dh > 64 ? launch_mfma<128>(a, st) : launch_mfma<64>(a, st);
    }
    switch (dh) {
        case 4: return launch_small<4>(a, st);
        case 8: return launch_small<8>(a, st);
        case 16: return launch_small<16>(a, st);
        case 32: return launch_small<32>(a, st);
        default: break;
    }
    if (dh > 256) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_attention_f32: head dim %d not supported (<= 256)", dh);
    const int total = nseq * H * Tq;
    hipLaunchKernelGGL(attn_wave_kernel, dim3((total + 3) / 4), dim3(256), 0, st, a, dh);
    return mmdm_check_launch("attn_wave");
}

extern "C" int mmdm_attention_planes(const void* Qp, int ldq, int64_t q_plane, const void* Kp, int ldk, int64_t k_plane, int nplanes, const float* V, int ldv,
                                     void* Ov, int ldo, int out_mode, int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    if (nplanes == 2) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: the two-plane fp16 form is mmdm_attention_split (V as planes too)");
    return mmdm_attention_planes_ex(Qp, ldq, q_plane, Kp, ldk, k_plane, nplanes, V, ldv, nullptr, 0, 0, Ov, ldo, out_mode, flags, nseq, Tq, Tk, H, dh, kv_seq_shift, stream);
}

extern "C" int mmdm_attention_bf16(const void* Qp, int ldq, const void* Kp, int ldk, const void* Vp, int ldv, void* O, int ldo, int out_mode, int flags,
                                   int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    if (!Vp) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_bf16: V is null");
    return mmdm_attention_planes_ex(Qp, ldq, 8, Kp, ldk, 8, 1, nullptr, 0, Vp, ldv, 0, O, ldo, out_mode, flags, nseq, Tq, Tk, H, dh, kv_seq_shift, stream);     // no fp32 V
}

extern "C" int mmdm_attention_split(const void* Qp, int ldq, int64_t q_plane, const void* Kp, int ldk, int64_t k_plane, const void* Vp, int ldv, int64_t v_plane,
                                    void* O, int ldo, int out_mode, int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    if (!Vp) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_split: V is null");
    return mmdm_attention_planes_ex(Qp, ldq, q_plane, Kp, ldk, k_plane, 2, nullptr, 0, Vp, ldv, v_plane, O, ldo, out_mode, flags, nseq, Tq, Tk, H, dh, kv_seq_shift, stream);
}

// Vp != nullptr, one plane: V also as bf16 rows [rows][ldvp] -> P.V on the bf16 matrix cores (attn_qkp_kernel<DH, 1, true>);
// nplanes == 2: Q, K and V (Vp, plane stride v_plane) as the two fp16 planes of the fp32-split mode (attn_qkp_kernel<DH, 2, true, true>)
int mmdm_attention_planes_ex(const void* Qp, int ldq, int64_t q_plane, const void* Kp, int ldk, int64_t k_plane, int nplanes, const float* V, int ldv,
                             const void* Vp, int ldvp, int64_t v_plane, void* Ov, int ldo, int out_mode, int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream,
                             const mmdm_rag_seq* rg) {
    if (nseq == 0 || Tq == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    if (!Qp || !Kp || (!V && !Vp) || !Ov || nseq < 0 || Tq < 0 || Tk <= 0 || H <= 0 || (nplanes < 1 || nplanes > 3) || (nplanes == 2 && !Vp))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: bad arguments nseq=%d Tq=%d Tk=%d H=%d planes=%d", nseq, Tq, Tk, H, nplanes);
    if (dh != 64 && dh != 128) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_attention_planes: head dim %d not supported (64, 128)", dh);
    if (flags & ~(MMDM_ATTN_NO_ZERO_KEY | MMDM_ATTN_CAUSAL)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: unknown flags 0x%x", flags);
    if ((flags & MMDM_ATTN_CAUSAL) && (Tq != Tk || !(flags & MMDM_ATTN_NO_ZERO_KEY)))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: the causal mask needs Tq == Tk and no zero key");
    if (ldq < H * dh || ldk < H * dh || (!Vp && ldv < H * dh) || ldo < H * dh) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: row strides must cover H*dh=%d", H * dh);
    // the fp32 V is not read when bf16 V rows are given (all-bf16 attention)
    const bool al = ((reinterpret_cast<uintptr_t>(Qp) | reinterpret_cast<uintptr_t>(Kp) | (Vp ? 0 : reinterpret_cast<uintptr_t>(V))) & 15) == 0 && ((ldq | ldk) & 7) == 0 &&
                    (Vp || (ldv & 3) == 0) && (q_plane & 7) == 0 && (k_plane & 7) == 0;
    if (!al) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: Q/K planes and V must be 16-byte aligned (bf16 strides %% 8, fp32 strides %% 4)");
    if ((reinterpret_cast<uintptr_t>(Ov) & 15) || (ldo & 3)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: O must be 16-byte aligned with a row stride %% 4 == 0");
    AttnArgs a;
    a.Q = nullptr; a.K = nullptr; a.V = V; a.O = static_cast<float*>(Ov); a.ldq = 0; a.ldk = 0; a.ldv = ldv; a.ldo = ldo;
    a.Qp = static_cast<const __bf16*>(Qp); a.Kp = static_cast<const __bf16*>(Kp); a.q_plane = (size_t)q_plane; a.k_plane = (size_t)k_plane; a.ldqp = ldq; a.ldkp = ldk;
    a.Vp = static_cast<const __bf16*>(Vp); a.ldvp = ldvp; a.v_plane = (size_t)v_plane;
    if (Vp && (nplanes == 3 || (reinterpret_cast<uintptr_t>(Vp) & 15) || (ldvp & 7) || ldvp < H * dh || (nplanes == 2 && (v_plane & 7))))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_planes: 16-bit V needs one (bf16) or two (fp16 split) planes, 16-byte aligned rows / planes and a row stride >= H*dh");
    a.nseq = nseq; a.Tq = Tq; a.Tk = Tk; a.H = H; a.out_bf16 = out_mode; a.flags = flags; a.dh = dh; a.ablate = g_attn_ablate; a.stamps = g_attn_stamps;
    a.seq_off = rg ? rg->off : nullptr; a.seq_len = rg ? rg->len : nullptr;
    a.seq_order = (rg && rg->order && rg->items > 0 && nseq % rg->items == 0) ? rg->order : nullptr; a.order_items = rg ? rg->items : 0;
    a.o_plane = (size_t)(rg ? (size_t)rg->total_rows : (size_t)nseq * Tq) * ldo;
    if (rg && flags) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "attention on ragged batches: zero key, no mask (flags=0x%x)", flags);
    a.shift = ((kv_seq_shift % nseq) + nseq) % nseq;
    a.qtiles = (Tq + QB - 1) / QB;
    a.pairs_per_xcd = (nseq * H + 7) / 8;
    a.scale = 1.0f / sqrtf((float)dh);
    a.scale2 = a.scale * 1.4426950408889634f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nplanes == 3) return dh == 128 ? launch_qkp<128, 3>(a, st) : launch_qkp<64, 3>(a, st);
    if (nplanes == 2) return dh == 128 ? launch_qkp<128, 2, true, true>(a, st) : launch_qkp<64, 2, true, true>(a, st);
    if (Vp && g_attn_kc32) return dh == 128 ? launch_b32<128>(a, st) : launch_b32<64>(a, st);
    if (Vp) return dh == 128 ? launch_qkp<128, 1, true>(a, st) : launch_qkp<64, 1, true>(a, st);
    return dh == 128 ? launch_qkp<128, 1>(a, st) : launch_qkp<64, 1>(a, st);
}

// diagnostics of this translation unit (mmdm_diag_set): ablation bits, in-kernel stamp buffer
bool mmdm_diag_attn(const char* key, long long v) {
    if (!strcmp(key, "attn_ablate")) g_attn_ablate = (int)v;
    else if (!strcmp(key, "attn_kc32")) g_attn_kc32 = (int)v;
    else if (!strcmp(key, "attn_stamps")) g_attn_stamps = reinterpret_cast<unsigned long long*>((uintptr_t)v);
    else return false;
    return true;
}
