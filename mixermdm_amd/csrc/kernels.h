// Internal helpers shared by the HIP translation units of libmmdm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmdm.h"

// printf-style: records the message for mmdm_last_error() and returns `code`.
int mmdm_set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// hipGetLastError() after a launch; returns MMDM_OK or MMDM_ERR_HIP (message recorded).
int mmdm_check_launch(const char* what);
// One-time per-process kernel attribute setup (dynamic LDS sizes); safe to call repeatedly, never during capture.
int mmdm_kernels_init(void);
// Records which GEMM instantiation the last mmdm_linear_* call of this thread launched (mmdm_last_gemm_kernel(), a debug getter the
// parity tests use to assert that a shape really lands on the production tiles).  A call that splits its rows over two kernels
// records both, "+"-joined.
void mmdm_note_gemm(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void mmdm_note_gemm_reset(void);
int mmdm_gemm_init(void);
// Row split of a GEMM's fractional last round of tiles (gemm_f32.hip): t/10 = the largest fraction of a round that is split off; 0 = never.
// Thread-local; the sampler sets it per handle before it launches a step (one-stream samplers benefit, the two-stream step does not).
void mmdm_gemm_set_tail(int t);
int mmdm_gemm_get_tail(void);
int mmdm_gemm_bf16_init(void);
int mmdm_gemm_split_init(void);
// per-translation-unit halves of mmdm_diag_set (include/mmdm.h section 4): true if `key` belongs to the unit
bool mmdm_diag_gemm_f32(const char* key, long long v);
bool mmdm_diag_gemm_bf16(const char* key, long long v);
bool mmdm_diag_gemm_split(const char* key, long long v);
bool mmdm_diag_attn(const char* key, long long v);

// Activations of the GEMM epilogues, one definition for every kernel.  Each is a FIXED sequence of operations (explicit fma, no
// expression the compiler may or may not contract), so two instantiations of an epilogue -- fp32 rows vs bf16 planes, one tile shape vs
// another -- produce the same bits for the same accumulator value.
// erf(a), branch-free, <= 1 ulp on each of its two ranges (minimax polynomials: N. Juffa's single-precision erff, published under the
// BSD 2-clause licence): both ranges are evaluated and one is selected, so a wave never diverges -- the library erff branches on |a|,
// and the 32-64 copies of it in a GEMM epilogue are what that epilogue spends its time in (tools/gemm_timeline.py: 39 us per 128 x 64
// tile beside co-resident workgroups in their K loops, 8.6 us with the activation removed).
__device__ __forceinline__ float erf_bf(float a) {
    const float t = __builtin_fabsf(a), s = a * a;
    // |a| > 0.927734375:  1 - exp(p(t))
    float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    r = 1.0f - __builtin_amdgcn_exp2f(r * 1.44269504088896340736f);
    r = __builtin_copysignf(r, a);
    // |a| <= 0.927734375:  a + a q(a^2)
    float q = -5.96761703e-4f;
    q = __builtin_fmaf(q, s, 4.99119423e-3f);
    q = __builtin_fmaf(q, s, -2.67681349e-2f);
    q = __builtin_fmaf(q, s, 1.12819925e-1f);
    q = __builtin_fmaf(q, s, -3.76125336e-1f);
    q = __builtin_fmaf(q, s, 1.28379166e-1f);
    q = __builtin_fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}
__device__ __forceinline__ float gelu_erf(float x) {            // F.gelu (erf form): 0.5 x (1 + erf(x / sqrt 2))   src/models/utils/layers.py:104
    const float h = 0.5f * x;
    return __builtin_fmaf(h, erf_bf(x * 0.70710678118654752440f), h);
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float quick_gelu(float x) { return x * (1.0f / (1.0f + expf(-1.702f * x))); }   // CLIP QuickGELU: x * sigmoid(1.702 x)

// Operand format of the fp32-split precision mode (gemm_split.hip): x ~= h + l / 2048 with h = fp16(x), l = fp16((x - h) * 2048), both
// round-to-nearest -- 11 + 11 significand bits, |x - (h + l/2048)| <= 2^-22 |x| (down to |x| = 2^-14; below that the absolute error stays under
// 2^-36), the second plane carried at 2^11 so that it never meets fp16's subnormals.  |x| must stay below 65504 (inf and then NaN otherwise:
// loud, not wrong).  One definition for every producer (AdaLN, attention, the GELU epilogue, the weight conversion).
constexpr int MMDM_SPLIT_NPL = 2;
constexpr float MMDM_SPLIT_SCALE = 2048.0f, MMDM_SPLIT_INV = 1.0f / 2048.0f;
typedef _Float16 mmdm_h4 __attribute__((ext_vector_type(4)));
// (the empty asm keeps x a materialised fp32 value: without it the compiler may fuse the producer's last fma with this conversion into
//  v_fma_mixlo_f16 -- ONE rounding of the exact fma result to fp16 instead of two -- and the hi plane of an epilogue then differs at exact fp16
//  ties from the split of the same epilogue's fp32 output; seen once gemm_split.hip was built without packed-fp32 instructions, round 6)
__device__ __forceinline__ _Float16 mmdm_split_hi(float x) { asm volatile("" : "+v"(x)); return (_Float16)x; }
__device__ __forceinline__ _Float16 mmdm_split_lo(float x, _Float16 h) { return (_Float16)((x - (float)h) * MMDM_SPLIT_SCALE); }
__device__ __forceinline__ void mmdm_split2(float x, _Float16& h, _Float16& l) { h = mmdm_split_hi(x); l = mmdm_split_lo(x, h); }
__device__ __forceinline__ void mmdm_split2(const float (&x)[4], mmdm_h4& h, mmdm_h4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const _Float16 t = mmdm_split_hi(x[e]); h[e] = t; l[e] = mmdm_split_lo(x[e], t); }
}

constexpr int MMDM_NF = 262;      // pose features per person (src/models/in2in.py:426, INPUT_DIM)
constexpr int MMDM_NJ = 22;       // joints

int mmdm_attn_init(void);
int mmdm_step_dec(int* step_idx, int* loop_pos, hipStream_t st);
int mmdm_set_step(int* step_idx, int* loop_pos, int s, int l, hipStream_t st);
int mmdm_gather_rows(const float* src, const int* idx, float* dst, int n, int D, hipStream_t st);
// History destinations of the current sampling call, kept in DEVICE memory and read by the step's kernels at run time, so that a
// captured step graph is independent of them (mmdm_set_history rewrites the descriptor, never the graph).
struct mmdm_hist_desc {
    float *i1, *i2, *o1, *o2, *mix;   // influence_i1/i2 [slots, 2B, T, 262 or 1], out1/out2/out_influenced [slots, 2B, T, 524]; null = not kept
    int every;                        // slot k receives the step at loop position k*every
    int pad;
};
int mmdm_set_hist_desc(mmdm_hist_desc* d, const mmdm_hist_desc& v, hipStream_t st);
int mmdm_hist_copy(const float* src, const mmdm_hist_desc* hd, int which, size_t count, const int* loop_pos, hipStream_t st);
int mmdm_blend_cfg_dyn(const float* out1, const float* out2, const float* w, int mode, int use_force, float force, float cfg_scale,
                       float* model_out, const mmdm_hist_desc* hd, const int* loop_pos, int B, int T, hipStream_t st);
// A RAGGED batch as the geometry kernels see it (device arrays written by mmdm_rag_setup; one group = the B items' frames back to back, padded to `rows`)
constexpr int MMDM_RAG_MAX_ITEMS = 256;      // item lengths travel to the device by value (kernel argument)
struct mmdm_rag {
    const int* row_item;   // [rows] item of a frame row; -1 = padding row
    const int* row_pos;    // [rows] frame index inside its item
    const int* item_off;   // [B] first frame row of an item
    const int* item_len;   // [B]
    int B, rows;           // items; group stride in frame rows (sum of lengths rounded up to the handle's row bucket)
};
int mmdm_rag_setup(const int* lens_host, int B, int rows, int groups, int* item_off, int* item_len, int* row_item, int* row_pos, int* row_seq,
                   int* seq_off, int* seq_len, int* item_order, hipStream_t st);
int mmdm_mixer_pre_rag(const float* o1, const float* o2, const float* stats, float* out1, float* out2, int groups, int align, const mmdm_rag& rg, hipStream_t st);
int mmdm_blend_cfg_rag(const float* out1, const float* out2, const float* w, int mode, int use_force, float force, float cfg_scale,
                       float* model_out, const mmdm_hist_desc* hd, const int* loop_pos, const mmdm_rag& rg, hipStream_t st);
int mmdm_xstart_ddim_rag(const float* model_out, const float* stats, const float* coef, int S, const int* step_idx,
                         float* x, float* x2, float* pred_xstart, float* pred_xstart2, float* floor_ws, int align, const mmdm_rag& rg, hipStream_t st);
int mmdm_linear_f32_ex(const float* A, int lda, const float* W, int ldw, int Kw, const float* bias, float* C, int ldc,
                       int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
int mmdm_mdm_pack(const float* src, const float* cond, int ldc, const float* time_tab, const int* step_idx, const float* pe, float* dst,
                  int nseq, int T, int D, hipStream_t st);
int mmdm_mdm_unpack(const float* src, float* dst, int nseq, int T, int D, hipStream_t st);
int mmdm_repack_pose(const float* src, int ld_src, float* dst, int npers, int rows, int ldp, int split, hipStream_t st);
int mmdm_linear_split_ex(const void* A, int lda, int64_t a_plane, const void* W, int ldw, int64_t w_plane, const float* bias, void* C, int ldc,
                         int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period,
                         void* planes2, int ld2, int64_t plane2_stride, int planes2_cols, void* stream);
int mmdm_linear_bf16_ex(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int out_bf16,
                        int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* bf16_copy, int ld2, int copy_cols, void* stream);
int mmdm_linear_fp8_ex(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* C, int ldc,
                       int out_mode, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* bf16_copy, int ld2, int copy_cols,
                       float a_const, float out_scale, void* stream);
int mmdm_adaln_any(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, int out_mode, float* row_scale, int nseq, int T, int D, void* stream,
                   const int* row_seq = nullptr, int rows_rag = 0);
int mmdm_mean_time_rag(const float* h, float* out, int nseq, const int* seq_off, const int* seq_len, int D, hipStream_t st);
// rowops.hip's second build (rowops_nopk.o: no packed-fp32 VALU instructions; mixermdm_amd/build.py), taken by precision 1-3 handles through mmdm.hip's ROWOP
int mmdm_adaln_any_nopk(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, int out_mode, float* row_scale, int nseq, int T, int D, void* stream,
                        const int* row_seq = nullptr, int rows_rag = 0);
int mmdm_mean_time_rag_nopk(const float* h, float* out, int nseq, const int* seq_off, const int* seq_len, int D, hipStream_t st);
int mmdm_mdm_pack_nopk(const float* src, const float* cond, int ldc, const float* time_tab, const int* step_idx, const float* pe, float* dst,
                       int nseq, int T, int D, hipStream_t st);
int mmdm_mdm_unpack_nopk(const float* src, float* dst, int nseq, int T, int D, hipStream_t st);
// silu(time_tab[*step_idx] + txt) as the two fp16 planes of the fp32-split GEMM's A operand (rowops.hip)
int mmdm_cond_silu_planes(const float* time_tab, const int* step_idx, const float* txt, _Float16* out, size_t plane, int rows, int D, hipStream_t st);
int mmdm_cond_silu_planes_nopk(const float* time_tab, const int* step_idx, const float* txt, _Float16* out, size_t plane, int rows, int D, hipStream_t st);
extern "C" {
int mmdm_layernorm_f32_nopk(const float* x, const float* gamma, const float* beta, float* out, int rows, int D, float eps, void* stream);
int mmdm_cond_silu_f32_nopk(const float* time_tab, const int* step_idx, const float* txt, float* out, int rows, int D, void* stream);
int mmdm_mean_time_f32_nopk(const float* h, float* out, int nseq, int T, int D, void* stream);
}
// Sequences of a RAGGED batch as the attention kernels see them (device arrays: one captured graph serves every batch of the same row bucket):
// sequence s owns rows [off[s], off[s] + len[s]) of Q / K / V / O; total_rows = rows of O (plane stride of a split output); max_len sizes the grid.
// order / items (optional): the batch's B items sorted by length, longest first (written by mmdm_rag_setup); the nseq = k x B sequences of a launch are then
// dealt to the workgroups longest first -- workgroup slot j / H handles sequence (j' % k) * B + order[j' / k] -- so that the launch does not end on a 300-frame
// sequence that started late (a pure re-numbering of who computes what: every sequence's arithmetic is untouched).  nullptr: sequences in index order.
struct mmdm_rag_seq { const int* off; const int* len; int total_rows; int max_len; const int* order = nullptr; int items = 0; };
int mmdm_attention_planes_ex(const void* Qp, int ldq, int64_t q_plane, const void* Kp, int ldk, int64_t k_plane, int nplanes, const float* V, int ldv,
                             const void* Vp, int ldvp, int64_t v_plane, void* O, int ldo, int out_mode, int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream,
                             const mmdm_rag_seq* rg = nullptr);
int mmdm_attention_opts_rag(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* O, int ldo, int out_bf16,
                            int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, const mmdm_rag_seq* rg, void* stream);
