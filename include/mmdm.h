/* mmdm.h -- C ABI of libmmdm_hip.so: the MI355X (gfx950) implementation of MixerMDM's denoising hot path.
 *
 * The reference (pabloruizponce/MixerMDM) has no FFI layer: its boundary for this path is a Python
 * nn.Module protocol.  This header is what a binding of that protocol calls; each entry
 * point names the reference code it replaces (paths relative to /root/reference).  INTEGRATION.md shows the
 * ctypes stub that wires these into the reference's `src/models/mixermdm.py`.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless it is documented as host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); no entry point synchronises the host except mmdm_destroy
 *     (which waits for the device) and mmdm_run when its graph cache has to evict an entry whose replays are still queued;
 *   - the callee never frees or keeps caller memory (weights are copied into the handle's packed layout);
 *   - return value: 0 = MMDM_OK, otherwise an mmdm_status; mmdm_last_error() gives the message
 *     (thread-local for the stateless kernels, per handle otherwise);
 *   - a handle is not thread-safe and allocates nothing after mmdm_prepare() (graph-capturable); several handles per device may exist and
 *     be used from ONE host thread (mmdm_create_shared lets them share a weight set; see there for what overlaps).
 *
 * Environment variables the library reads (all optional; nothing else in the environment changes its behaviour):
 *   MMDM_NO_OVERLAP=1    mmdm_create: run the two denoisers and the two Influence calls of a step on ONE stream (profiling passes:
 *                        a kernel trace without concurrent kernels); results are bit-identical either way.
 *   MMDM_NO_SPLIT_EMBED=1  precision 1-3: keep motion_embed on the fp32 MFMA kernel instead of the fp32-split kernel (A/B timing and accuracy)
 *   MMDM_NO_SPLIT_COND=1   precision 1-3: keep the AdaLN conditioning projections (silu(time + text) against [L*n_ada*2D, D]) on the fp32 MFMA kernel
 *                        instead of the fp32-split kernel (A/B timing and accuracy)
 *   MMDM_GRAPH_CACHE=n   mmdm_create: capacity (1..64, default 8) of the handle's (B, T, S)-keyed cache of captured step graphs.
 *   MMDM_RAG_BUCKET=n    mmdm_create: row granularity (1..4096, default 128) to which a ragged call's group of frames is padded (mmdm_begin_ragged).
 *   MMDM_SERIALIZE_HANDLES=1  the first mmdm_create of the process: every sampling call (mmdm_run) waits ON THE DEVICE for the previous sampling call
 *                        of any handle of the process (one process-wide event; no host synchronisation).  Off by default: handles of every
 *                        precision overlap bit-exactly (tests/test_gpu_ragged.py); the switch exists so that overlap can be ruled out in the field.
 *   MMDM_NO_PACK=1       keep the low-precision weight twins of precision 1-3 in row-major planes instead of MFMA fragment order (the
 *                        packed and the plane kernels are bit-identical; tests/test_gpu_packed_modes.py compares them).
 *   MMDM_QKP / MMDM_NO_QKP / MMDM_NO_BF16_PV   precision >= 1: force / forbid the bf16-plane Q K^T and the bf16 P V forms of the attention.
 * All of them are read ONCE, by mmdm_create, into the handle: a handle's behaviour never changes after it exists, and the stateless kernels
 * of section 1 read no environment at all.  The library exports exactly the symbols this header declares (hidden visibility otherwise);
 * section 4 is the one diagnostic entry point the scripts under tools/ use.
 */
#ifndef MMDM_H
#define MMDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)      /* the library is built with -fvisibility=hidden: only what is declared here is exported */

typedef enum {
    MMDM_OK = 0,
    MMDM_ERR_ARG = 1,      /* bad shape / alignment / enum */
    MMDM_ERR_STATE = 2,    /* missing weight, prepare() not called, ... */
    MMDM_ERR_HIP = 3,      /* a HIP runtime call or kernel launch failed */
    MMDM_ERR_UNSUPPORTED = 4
} mmdm_status;

const char* mmdm_last_error(void);
/* "gfx950;<build id>" -- lets the host check that the right library was loaded. */
const char* mmdm_version(void);
/* Debug getter: the GEMM instantiation(s) the calling thread's last mmdm_linear_f32 / _bf16 / _split / _fp8 call launched, e.g.
 * "gemm_glds<22,22,16,2,vepi>" (tile code: 10*waves + 32x32-tiles-per-wave along M, N; K step; LDS buffers; epilogue form) or
 * "gemm_split<42,22>+gemm_split<22,21>" when a call splits its rows over two kernels.  Lets parity tests assert that a shape
 * really runs on the production tiles. */
const char* mmdm_last_gemm_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * 1. Stateless kernels (used by the handle below and exposed for per-kernel parity tests).
 * ---------------------------------------------------------------------------------------------- */

/* Epilogues of mmdm_linear_f32. */
enum {
    MMDM_EPI_BIAS = 0,       /* C = A W^T + b                                   nn.Linear */
    MMDM_EPI_BIAS_GELU = 1,  /* C = gelu_erf(A W^T + b)                         FFN.linear1+activation  src/models/utils/layers.py:104 */
    MMDM_EPI_BIAS_RESID = 2, /* C = A W^T + b + R   (R may alias C)             "+ x" residuals         src/models/utils/blocks.py:50-63 */
    MMDM_EPI_BIAS_PE = 3,    /* C = A W^T + b + pe[m % period]                  motion_embed + PositionalEncoding  src/models/in2in.py:426-431 */
    MMDM_EPI_BIAS_SILU = 4,  /* C = silu(A W^T + b)                             TimestepEmbedder time_embed.0+SiLU src/models/utils/utils.py:47-51 */
    MMDM_EPI_BIAS_QUICKGELU = 5, /* C = z*sigmoid(1.702 z), z = A W^T + b           CLIP text tower MLP (clip==1.0 model.py QuickGELU; src/models/mixermdm.py:213) */
    MMDM_EPI_BIAS_SIGMOID = 6   /* C = sigmoid(A W^T + b)                           Influence.out + sigmoid  src/models/utils/influence.py:124-125 */
};

/* y = x W^T + b with a fused epilogue; exact fp32 (v_mfma_f32_32x32x2_f32).
 * A [M,K] row stride lda; W [N,K] row stride ldw (nn.Linear layout); bias [N] or NULL; C [M,N] row stride ldc;
 * extra: R [M,N] row stride ld_extra (RESID) or pe table [period, N] row stride ld_extra (PE).
 * Replaces torch.nn.functional.linear at every call site on the path (SURVEY.md 2.3 K1,K2,K4,K6,K7,K8). */
int mmdm_linear_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                    int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);

/* bf16-operand variant (BASELINE configs[4]): A [M,K] and W [N,K] are bf16 (16-byte aligned rows, K % 32 == 0), accumulation
 * fp32 (v_mfma_f32_32x32x16_bf16), bias / residual / PE fp32, C fp32 (out_bf16 = 0) or bf16 (out_bf16 = 1); N % 4 == 0. */
int mmdm_linear_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int out_bf16,
                     int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
/* Static weights in MFMA FRAGMENT ORDER for the bf16 / fp8 linear layers: mmdm_pack_weight_frag permutes W [N][row_bytes] (bf16: row_bytes = 2 K,
 * fp8: K) inside blocks of 32 rows x 32 bytes so that one wave-wide 16-byte load is one MFMA operand; the *_packed entry points then take W
 * straight from global memory (LDS carries A only; 128 x 256 or 128 x 128 tiles for bf16, 128 x 128 for fp8, K step 128 bytes).  Bit-identical to
 * mmdm_linear_bf16 / mmdm_linear_fp8.  Needs N % 128 == 0 and K % 128 == 0 (bf16) / K % 256 == 0 (fp8); a row slice starting at a multiple of 32 rows is the same byte offset. */
int mmdm_pack_weight_frag(const void* W, int64_t ld_bytes, void* out, int N, int row_bytes, void* stream);
int mmdm_linear_bf16_packed(const void* A, int lda, const void* W_packed, const float* bias, void* C, int ldc, int out_bf16,
                            int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
int mmdm_linear_fp8_packed(const void* A, int lda, const float* a_scale, const void* W_packed, const float* w_scale, const float* bias, void* C, int ldc,
                           int out_mode, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
/* Round-to-nearest-even fp32 -> bf16 conversion of n contiguous elements. */
int mmdm_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);

/* fp8 operands (OCP e4m3; BASELINE configs[4] "fp8 MFMA QKV/FFN GEMMs", mmdm_config.precision = 3): A [M,K] and W [N,K] are fp8 bytes
 * (16-byte aligned rows, K % 64 == 0), accumulated in fp32 on v_mfma_scale_f32_32x32x64_f8f6f4 (block-scaled form, unit scales) and de-quantised in the epilogue:
 *   C[m][n] = acc[m][n] * a_scale[m] * w_scale[n] + bias[n] (+ residual / PE row), then the activation;
 * a_scale [M] per-row activation scales, w_scale [N] per-output-channel weight scales (either may be NULL = 1).
 * out_mode: 0 fp32, 1 bf16, 2 fp8 at unit scale (values saturate at +-448).  Epilogues / extra / period as mmdm_linear_f32; N % 4 == 0. */
int mmdm_linear_fp8(const void* A, int lda, const float* a_scale, const void* W, int ldw, const float* w_scale, const float* bias, void* C, int ldc,
                    int out_mode, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
/* Row-wise e4m3 quantisation of in [rows, K] (fp32, row stride ld_in): scale[r] = max|in[r,:]| / 448 (1 for a zero row),
 * out[r,k] = e4m3(in[r,k] / scale[r]), round to nearest even.  Per-output-channel weight quantisation = this on W [N,K]. */
int mmdm_quantize_rows_fp8(const float* in, int ld_in, void* out, int ld_out, float* scale, int rows, int K, void* stream);
/* AdaLN apply (mmdm_adaln_f32) writing the fp8 GEMM's A operand directly: out [rows, D] e4m3 bytes + row_scale [rows]. */
int mmdm_adaln_fp8(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, float* row_scale, int nseq, int T, int D, void* stream);

/* fp32 linear layer on the 16-bit matrix cores from two-way fp16 operand splits ("fp32-split" precision, mmdm_config.precision = 2):
 * every fp32 x is carried as h = fp16(x), l = fp16((x - h) * 2048), x ~= h + l / 2048 (11 + 11 significand bits: relative error <= 2^-22,
 * |x| < 65504); a*w is accumulated in fp32 as  hi += ah*wh,  lo += ah*wl + al*wh,  result = hi + lo / 2048  (three
 * v_mfma_f32_32x32x16_f16; the dropped al*wl is < 2^-22 |a||w|): as accurate against a float64 product as the fp32 MFMA kernel (whose
 * accumulation rounding over K >= 256 terms is the larger error) at 3/16 of its matrix-core cost.  A and W are given as two fp16 planes
 * [2][rows][K] (plane strides in elements, from mmdm_f32_split or a split-writing producer); C is fp32 [M,N], or its two fp16 planes if
 * out_split.  Epilogues / extra / period as mmdm_linear_f32; K % 32 == 0, N % 4 == 0.  (Rounds 1-3 used an exact three-way bf16 split with
 * six MFMAs per product block; measured no more accurate, twice the matrix-core work: LAB_NOTES.md, round 4.) */
int mmdm_linear_split(const void* A, int lda, int64_t a_plane, const void* W, int ldw, int64_t w_plane, const float* bias, void* C, int ldc,
                      int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
/* The same product with W in FRAGMENT ORDER (static weights): mmdm_split_pack_weight permutes the two planes [2][N][K] inside blocks
 * of 32 rows x 16 k so that one wave-wide 16-byte load is one MFMA operand; the kernel then takes W straight from global memory and only
 * A goes through LDS.  Same term order per accumulator: results are bit-identical to mmdm_linear_split.  Needs N % 64 == 0, K % 64 == 0
 * (packing alone: N % 32 == 0, K % 16 == 0); a row slice that starts at a multiple of 32 rows is the same offset as in the plane layout. */
int mmdm_linear_split_packed(const void* A, int lda, int64_t a_plane, const void* W_packed, int64_t w_plane, const float* bias, void* C, int ldc,
                             int64_t c_plane, int out_split, int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream);
int mmdm_split_pack_weight(const void* W, int ldw, int64_t w_plane, void* out, int64_t out_plane, int N, int K, void* stream);
/* Split of n contiguous fp32 values into the two fp16 planes out[0], out[plane_stride] (elements): in ~= out0 + out1 / 2048. */
int mmdm_f32_split(const float* in, void* out, int64_t n, int64_t plane_stride, void* stream);

/* AdaLN apply: out[s,t,:] = LN_{eps=1e-6,no affine}(h[s,t,:]) * (1 + ss[row(s), 0:D]) + ss[row(s), D:2D],
 * row(s) = s % ss_rows.  ss is the output of Linear(SiLU(emb)) (scale first, shift second), row stride ss_ld.
 * Replaces AdaLN.forward  src/models/utils/layers.py:15-25. */
int mmdm_adaln_f32(const float* h, const float* ss, int ss_ld, int ss_rows, float* out, int nseq, int T, int D, void* stream);

/* Same with a selectable output type: out_bf16 != 0 writes `out` as bf16 (operand of the next bf16 GEMM). */
int mmdm_adaln_ex(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, int out_bf16, int nseq, int T, int D, void* stream);

/* Multi-head attention core with add_zero_attn (one extra key, logit 0, value 0), no masks, scale 1/sqrt(dh).
 * Q/K/V/O are [nseq, T*, H*dh] views with row strides ld*; the K/V sequence for query sequence s is
 * (s + kv_seq_shift) % nseq (used by the interaction denoiser: person a attends to person b's keys).
 * Replaces the SDPA inside nn.MultiheadAttention  src/models/utils/layers.py:33-44, 74-87. */
int mmdm_attention_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                       int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream);

/* Same with a selectable output type (fp32 Q/K/V in, fp32 softmax and accumulation; O fp32 or bf16). */
int mmdm_attention_ex(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* O, int ldo, int out_bf16,
                      int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream);

/* mmdm_attention_f32 over a RAGGED batch: sequence s owns rows [seq_off[s], seq_off[s] + seq_len[s]) of Q / K / V / O (DEVICE int arrays of nseq
 * entries; the K/V sequence of s is (s + kv_seq_shift) % nseq and may have another length); max_len >= every length sizes the grid; total_rows =
 * rows of O.  dh = 64 or 128.  Per sequence bit-identical to mmdm_attention_f32 on that sequence alone.  (The 16-bit forms have the same ragged
 * instantiations inside the sampler: mmdm_begin_ragged.) */
int mmdm_attention_ragged_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                              int nseq, const int* seq_off, const int* seq_len, int max_len, int total_rows, int H, int dh, int kv_seq_shift, void* stream);

/* Same with options.  flags: MMDM_ATTN_NO_ZERO_KEY = plain softmax over the Tk keys (nn.MultiheadAttention default, as inside
 * nn.TransformerEncoderLayer: MDMDenoiser.seqTransEncoder src/models/mdm.py:252-264, clipTransEncoder src/models/mixermdm.py:246-258);
 * MMDM_ATTN_CAUSAL = keys <= query only (the CLIP text tower's attention mask; needs NO_ZERO_KEY and Tq == Tk).
 * Head sizes 64 and 128 run on the MFMA kernel, 4..32 and any other size <= 256 on scalar fallbacks. */
enum { MMDM_ATTN_NO_ZERO_KEY = 1, MMDM_ATTN_CAUSAL = 2 };
int mmdm_attention_opts(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* O, int ldo, int out_bf16,
                        int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream);

/* Attention with Q K^T on the bf16 matrix cores: Q and K are given as `nplanes` bf16 planes [plane][rows][ld] (plane strides in elements).
 * nplanes = 3: exact 3-way bf16 splits of the fp32 projections (x = x1 + x2 + x3) -> fp32-accurate scores from six
 * v_mfma_f32_16x16x32_bf16 per block; nplanes = 1: bf16 Q and K (the bf16 path).  V is fp32; softmax, P.V and the output are those of
 * mmdm_attention_opts (out_mode: 0 fp32, 1 bf16, 2 the two fp16 planes of mmdm_linear_split; flags as there; zero key by default).  dh = 64 or 128. */
int mmdm_attention_planes(const void* Qp, int ldq, int64_t q_plane, const void* Kp, int ldk, int64_t k_plane, int nplanes, const float* V, int ldv,
                          void* O, int ldo, int out_mode, int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream);

/* The all-bf16 form (BASELINE configs[4] path): as mmdm_attention_planes with nplanes = 1 and V given as bf16 rows [rows][ldv] as well
 * (the projection GEMM's bf16 copy); P.V also runs on the bf16 matrix cores (v_mfma_f32_16x16x16_bf16, probabilities rounded to bf16,
 * V read transposed from its row-major LDS image by ds_read_b64_tr_b16), fp32 accumulation and softmax.  dh = 64 or 128. */
int mmdm_attention_bf16(const void* Qp, int ldq, const void* Kp, int ldk, const void* Vp, int ldv, void* O, int ldo, int out_mode, int flags,
                        int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream);

/* The attention of the fp32-split precision mode: Q, K and V are each given as the two fp16 planes of mmdm_linear_split (h, l with
 * x ~= h + l / 2048; [2][rows][ld], plane strides in elements) -- what the projection GEMM writes with out_split instead of fp32 rows.
 * Scores: hi += kh qh, lo += kl qh + kh ql (three v_mfma_f32_16x16x32_f16 per 32-deep step), S = hi + lo / 2048; fp32 softmax; the
 * probabilities are split the same way on the fly (kept <= 16 by the deferred running maximum) and P.V is three v_mfma_f32_16x16x16_f16 per 16 columns.
 * As accurate against float64 as mmdm_attention_f32 (tests), ~2.5x faster.  out_mode / flags as mmdm_attention_planes.  dh = 64 or 128. */
int mmdm_attention_split(const void* Qp, int ldq, int64_t q_plane, const void* Kp, int ldk, int64_t k_plane, const void* Vp, int ldv, int64_t v_plane,
                         void* O, int ldo, int out_mode, int flags, int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream);

/* out[r,:] = silu(time_row[:] + txt[r,:]) for r < rows; time_row = time_tab + (*step_idx) * D.
 * Replaces `embed_timestep(t) + text_embed(c)` followed by AdaLN's SiLU  in2in.py:415-422, layers.py:9-10. */
int mmdm_cond_silu_f32(const float* time_tab, const int* step_idx, const float* txt, float* out, int rows, int D, void* stream);

/* Mixer pre-processing of the two denoisers' outputs (CFG-doubled batch n = 2B):
 *   o1, o2 [n,T,524] (normalised) -> out1, out2 [n,T,524]: denormalise (HML3D stats for o1, InterHuman stats for o2),
 *   and if align != 0: ih_to_smpl, align_motions(target = o2 person, moved = o1 person), smpl_to_ih.
 * stats = [mean_hml | std_hml | mean_ih | std_ih], 4*262 floats.
 * Replaces src/models/mixermdm.py:691-719 + src/utils/alignment.py:11-158. */
int mmdm_mixer_pre_f32(const float* o1, const float* o2, const float* stats, float* out1, float* out2,
                       int n, int T, int align, void* stream);

/* Influence head: w[r, 0:nw] = sigmoid(h[r,:] Wout^T + b), nw = 1 or 23; h rows are [rows, D].
 * Replaces Influence.out + sigmoid  src/models/utils/influence.py:124-125. */
int mmdm_influence_head_f32(const float* h, const float* Wout, const float* bout, float* w, int rows, int D, int nw, void* stream);

/* Mean over time: out[s,:] = mean_t h[s,t,:]  (Influence modes 1 and 3, influence.py:120-121). */
int mmdm_mean_time_f32(const float* h, float* out, int nseq, int T, int D, void* stream);

/* Expand influence to 262 channels, blend, CFG-combine:
 *   mix = out2 + infl * (out1 - out2)   (per person);   model_out[b] = s*mix[b] + (1-s)*mix[B+b].
 * w: [2 persons, 2B, Tw, nw] where Tw = T (modes 2,4) or 1 (modes 1,3) and nw = 1 (modes 1,2) or 23 (modes 3,4);
 * force: if use_force != 0 every influence value is replaced by `force`.
 * hist_i1/hist_i2 [2B,T,262], hist_mix [2B,T,524]: optional side outputs (NULL to skip).
 * Replaces src/models/mixermdm.py:739-801 + ClassifierFreeSampleModelX2 combine  src/models/utils/cfg_sampler.py:49-55. */
int mmdm_blend_cfg_f32(const float* out1, const float* out2, const float* w, int mode, int use_force, float force,
                       float cfg_scale, float* model_out, float* hist_i1, float* hist_i2, float* hist_mix,
                       int B, int T, void* stream);

/* process_xstart + two-chain DDIM (eta = 0) update, in place on x and x2 [B,T,524]:
 *   i = *step_idx;  if i > 0: x0_1 = norm_hml(smpl_to_ih(center_motion(ih_to_smpl(m)))) per person (center only if align),
 *                             x0_2 = norm_ih(m);   else x0_1 = x0_2 = m        (m = model_out)
 *   eps = (coef[0][i]*x - x0) / coef[1][i];  x = x0*coef[2][i] + coef[3][i]*eps      (both chains)
 * coef: [4, S] fp32 = sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, sqrt(alphas_cumprod_prev), sqrt(1-alphas_cumprod_prev).
 * pred_xstart2 (optional) receives x0_2.  `floor_ws` is a [B*2] float scratch.
 * Replaces MixerDiffusion.p_mean_variance.process_xstart + ddim_sample  src/models/utils/gaussian_diffusion.py:2031-2062, 1936-1965
 * and center_motion  src/utils/alignment.py:161-222. */
int mmdm_xstart_ddim_f32(const float* model_out, const float* stats, const float* coef, int S, const int* step_idx,
                         float* x, float* x2, float* pred_xstart, float* pred_xstart2, float* floor_ws,
                         int B, int T, int align, void* stream);

/* Single-chain DDIM update (configs 1-2): x = x0*coef[2][i] + coef[3][i]*(coef[0][i]*x - x0)/coef[1][i], x0 = s*m[b] + (1-s)*m[B+b].
 * m is the denoiser output on the CFG-doubled batch [2B, T, C]; pred_xstart (optional) receives x0.
 * Replaces ClassifierFreeSampleModel combine + GaussianDiffusion.ddim_sample  cfg_sampler.py:24-28, gaussian_diffusion.py:799-849. */
int mmdm_cfg_ddim_f32(const float* m, const float* coef, int S, const int* step_idx, float cfg_scale,
                      float* x, float* pred_xstart, int B, int T, int C, void* stream);

/* Temporal smoothing of generated motions: out[b,t,c] = sum_k w[k+radius] * x[b, clamp(t+k, 0, T-1), c], accumulated in
 * double.  `weights` is a DEVICE array of 2*radius+1 doubles (the normalised gaussian of scipy's _gaussian_kernel1d).
 * Replaces scipy.ndimage.gaussian_filter1d(motion, 1, axis=0, mode='nearest')  src/scripts/infer/mixermdm.py:130. */
int mmdm_gaussian_filter1d_f32(const float* x, float* out, const double* weights, int radius, int n, int T, int C, void* stream);

/* 4-way CFG + single-chain DDIM: x0 = s*m[0:B] + s_int*m[B:2B] + s_ind*m[2B:3B] + (1-(s+s_int+s_ind))*m[3B:4B]; m [4B,T,C].
 * Replaces ClassifierFreeSampleModelMultiple combine + GaussianDiffusion.ddim_sample  cfg_sampler.py:90-98, gaussian_diffusion.py:799-849. */
int mmdm_cfg4_ddim_f32(const float* m, const float* coef, int S, const int* step_idx, float s, float s_int, float s_ind,
                       float* x, float* pred_xstart, int B, int T, int C, void* stream);

/* DualMDM composition + single-chain DDIM:  g_I = u_I + s_int (c_I - u_I),  g_i = u_i + s_ind (c_i - u_i)  (c = rows [0,B), u = rows
 * [B,2B) of the two models' outputs m_int / m_ind [2B,T,C]);  x0 = g_I + w_table[*step_idx] (g_i - g_I).
 * Replaces ClassifierFreeSampleDualMDM combine + GaussianDiffusion.ddim_sample  cfg_sampler.py:139-150, gaussian_diffusion.py:799-849. */
int mmdm_dual_ddim_f32(const float* m_ind, const float* m_int, const float* coef, int S, const int* step_idx, const float* w_table,
                       float s_ind, float s_int, float* x, float* pred_xstart, int B, int T, int C, void* stream);

/* nn.LayerNorm with affine: out[r,:] = (x[r,:] - mean) / sqrt(var + eps) * gamma + beta (biased variance); in place allowed.
 * Replaces norm1/norm2 of nn.TransformerEncoderLayer (src/models/mdm.py:252-264), clip_ln (src/models/mixermdm.py:259), ln_final. */
int mmdm_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out, int rows, int D, float eps, void* stream);

/* out[b,l,:] = table[tokens[b,l],:] + pos[l,:]; tokens: DEVICE int32 [n,L], clamped to [0, vocab).
 * Replaces token_embedding(text) + positional_embedding  src/models/mixermdm.py:298-299. */
int mmdm_token_embed_f32(const float* table, int vocab, const int* tokens, const float* pos, float* out, int n, int L, int D, void* stream);

/* dst[i,:] = src[idx[i],:]; idx: DEVICE int32 [n].  Replaces out[arange(B), text.argmax(-1)] (EOT row)  src/models/mixermdm.py:311. */
int mmdm_gather_rows_f32(const float* src, const int* idx, float* dst, int n, int D, void* stream);

/* One nn.TransformerEncoderLayer(batch_first=True) on x [nseq, T, D] in place, eval mode, dh = D/H.
 *   norm_first = 0 (torch default): x = LN1(x + SA(x)); x = LN2(x + W2 act(W1 x + b1) + b2)
 *       -- MDMDenoiser.seqTransEncoder (src/models/mdm.py:252-264), clipTransEncoder text heads (src/models/mixermdm.py:246-258, in2in.py:24-52)
 *   norm_first = 1: x += SA(LN1 x); x += W2 act(W1 LN2 x + b1) + b2 -- CLIP's ResidualAttentionBlock (clip==1.0 model.py; ln_1/ln_2 = norm1/norm2,
 *       mlp.c_fc/c_proj = linear1/linear2), with causal = 1 and activation = MMDM_EPI_BIAS_QUICKGELU
 * activation: MMDM_EPI_BIAS_GELU or MMDM_EPI_BIAS_QUICKGELU.  workspace: mmdm_encoder_layer_workspace(...) floats of device memory. */
typedef struct {
    const float *in_proj_weight, *in_proj_bias;     /* [3D, D], [3D]  (q, k, v order) */
    const float *out_proj_weight, *out_proj_bias;   /* [D, D], [D] */
    const float *linear1_weight, *linear1_bias;     /* [F, D], [F] */
    const float *linear2_weight, *linear2_bias;     /* [D, F], [D] */
    const float *norm1_weight, *norm1_bias, *norm2_weight, *norm2_bias;   /* [D] each */
} mmdm_encoder_layer_weights;
size_t mmdm_encoder_layer_workspace(int nseq, int T, int D, int F);
int mmdm_encoder_layer_f32(float* x, const mmdm_encoder_layer_weights* w, int nseq, int T, int D, int H, int F, int norm_first,
                           int activation, int causal, float eps, float* workspace, size_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 2. The sampler handle: weights + workspace + captured step graph.
 * ---------------------------------------------------------------------------------------------- */

typedef struct mmdm_handle_s* mmdm_handle;

typedef struct {
    /* denoisers (src/models/in2in.py:358-399; configs/models/{individual,in2IN}.yaml) */
    int d_latent, d_ff, d_layers, d_heads;
    /* mixer / Influence (src/models/mixermdm.py:606-657; configs/models/MixerMDM.yaml GENERATOR) */
    int m_latent, m_ff, m_layers, m_heads;
    int nfeats;        /* 262 */
    int text_dim;      /* 768 */
    int mixing_mode;   /* 1..4  (MIXING_MODE) */
    int align;         /* Mixer(align=...) */
    int xstart_align;  /* MixerDiffusion(align=...): the reference always leaves this True */
    int model2_kind;   /* 0 = in2IN interaction (three embs), 1 = InterGen InterDenoiser (one shared emb) */
    int use_force;     /* FORCE_INFLUENCE_VAL is not None */
    float force_val;
    float cfg_scale;   /* CFG_WEIGHT */
    int max_batch;     /* B (before CFG doubling) the workspace is sized for */
    int max_frames;    /* T */
    int single_only;   /* 0: two-chain MixerMDM.  1: single chain, denoiser1 only (individual in2IN or MDM, 2-way CFG; configs 1-2).
                        * 2: single chain, denoiser2 only (stand-alone interaction in2IN/InterGen, 4-way CFG of
                        *    ClassifierFreeSampleModelMultiple  src/models/utils/cfg_sampler.py:59-98; in2in.py:330-341)
                        * 3: single chain, in2IN "dual": denoiser1 in "dual_individual" and denoiser2 in "dual_interaction" mode composed by
                        *    ClassifierFreeSampleDualMDM  cfg_sampler.py:101-150, in2in.py:318-329; cond [B, 5*text_dim]; guidance scales
                        *    cfg_scale_individual / cfg_scale_interaction; needs mmdm_set_dual_weights */
    float cfg_scale_interaction, cfg_scale_individual;   /* CFG_WEIGHT_INTERACTION / CFG_WEIGHT_INDIVIDUAL (single_only == 2) */
    int precision;     /* 0: exact fp32 everywhere (the parity path).  1: bf16 operands for the transformer-stack GEMMs (weights
                        *    converted once at mmdm_prepare; AdaLN / attention / GELU outputs written as bf16), fp32 accumulation,
                        *    residual stream, softmax, geometry, DDIM and embeddings (BASELINE configs[4], "bf16 path")
                        * 2: "fp32-split": fp32 results from the 16-bit matrix cores -- the transformer-stack GEMM operands are two-way
                        *    fp16 splits (weights split at mmdm_prepare, AdaLN / attention / GELU outputs written as two planes) and each
                        *    product is accumulated as three fp16 MFMAs (mmdm_linear_split); accuracy = fp32 MFMA, everything else as 0
                        * 3: "bf16_fp8" (BASELINE configs[4]): as 1, with the QKV / cross-attention input projections and both FFN GEMMs on
                        *    fp8 e4m3 operands -- weights quantised per output channel at mmdm_prepare, AdaLN outputs quantised per row by
                        *    the AdaLN kernel, GELU outputs at unit scale -- fp32 accumulation and de-quantisation (mmdm_linear_fp8); the
                        *    attention output projections stay bf16
                        * RANGE PRECONDITION of precision 1, 2 and 3: some GEMM operands are carried in fp16 planes (precision 2: every operand of the
                        *    transformer stacks -- AdaLN outputs, Q|K|V, attention outputs, GELU hidden; precision 1, 2 and 3: the pose rows entering
                        *    motion_embed and the conditioning rows silu(time + text) entering the AdaLN projections, which run on the fp32-split
                        *    kernel), so those values must satisfy |x| < 65504.  A larger pose or activation
                        *    value becomes inf and then NaN in whole output rows (loud, never silently wrong); nothing clamps or checks at the boundary.
                        *    Poses in the models' normalised space are O(10); precision 0 has no such limit.  MMDM_NO_SPLIT_EMBED=1 / MMDM_NO_SPLIT_COND=1 keep
                        *    the embedding / the conditioning projections of precision 1 / 3 handles on the fp32 MFMA kernel (no range limit there). */
    int model1_kind;   /* 0 = in2IN individual denoiser, 1 = MDMDenoiser (post-norm nn.TransformerEncoder with a conditioning token,
                        *    src/models/mdm.py:234-298; MODEL1.NAME == "MDM", src/models/mixermdm.py:32-40, 264-265).  Its cond slices are
                        *    latent-sized (mdm.py:279), so the mixer's cond rows are [3*text_dim | 2*d1_latent | 3*text_dim] */
    int d1_latent, d1_ff, d1_layers, d1_heads;   /* denoiser1's own sizes (MODEL1 is a separate config); 0 = same as d_* */
} mmdm_config;

int mmdm_create(const mmdm_config* cfg, mmdm_handle* out);
/* A second sampler handle over the SAME weights: `parent` (prepared) keeps owning the parameter set, its low-precision twins and the normaliser
 * statistics; the new handle borrows them by reference (the block is freed when the last handle that holds it is destroyed, in any order) and
 * gets its own workspace sized for (max_batch, max_frames), its own streams, schedule tables, step state, history descriptor and graph cache.
 * Same configuration otherwise.  K such handles on K streams keep K independent sampling calls queued over one 1.46 GB weight copy.  What that
 * is good for: independent requests that arrive at different times share one weight copy.  What it is NOT: a throughput lever -- measured
 * (tools/inflight_probe.py, bench.py --eval-items: B = 1, fp32) the GPU co-schedules two such streams hardly at all, 1.00-1.03 x with 2-4 handles
 * (best case anywhere 1.22 x, eager with 16 hardware queues); packing the calls into one ragged batch (mmdm_begin_ragged) is what fills the
 * machine: 1.9 x (the reference's callers sample one item at a time: src/scripts/infer/mixermdm.py:184-188, src/evaluation/datasets.py:100-116).
 * A motion's bits do not depend on what runs beside it.  mmdm_set_weight / mmdm_set_norm_stats on a shared handle return MMDM_ERR_STATE (set
 * them on the parent; a parent re-prepared after mmdm_set_weight is seen by every holder); mmdm_prepare is a no-op there.  Handles are still
 * not thread-safe individually, and the handles of one process should be driven from ONE host thread: graph captures, instantiations,
 * evictions and replays are serialised against each other inside the library, but replaying graphs from two host threads was seen to crash
 * inside this runtime's hipGraphLaunch (ROCm 7.0 / 7.2, hip::Graph::UpdateStreams) -- every call here is asynchronous, one thread keeps K
 * streams queued.  Graph execs: destroying an exec beside another handle's live execs crashed that handle's next launch in this runtime, so
 * while handles share weights (and from then on in the process) a handle's graph cache does not evict: it grows to the number of DISTINCT
 * shapes the handle has sampled and every shape is captured once (mmdm_graph_parked counts the execs kept past their handle's life).
 * Sampling calls of different handles overlap on the device in every precision mode, bit-exactly (tests/test_gpu_ragged.py: every precision
 * pair; MMDM_SERIALIZE_HANDLES=1 serialises them).  What made two low-precision handles side by side give wrong motions for most of round 5
 * was the hardware, not the calls: on gfx950 dense VALU code with packed-fp32 instructions (v_pk_*_f32) in it transiently computes other bits
 * while its wave shares a SIMD with the packed-W GEMM kernels (root cause not isolated: LAB_NOTES.md); the library's geometry kernels, and
 * every row kernel of a precision 1-3 handle, are built without those instructions (build.py; tools/canary.hip is the stand-alone
 * reproducer).  A CALLER's own kernels that run beside a low-precision handle on the same device are exposed to the same hazard if they use
 * packed-fp32 arithmetic in bit-sensitive code (hipcc: -Xclang -target-feature -Xclang -packed-fp32-ops removes it). */
int mmdm_create_shared(mmdm_handle parent, int max_batch, int max_frames, mmdm_handle* out);
void mmdm_destroy(mmdm_handle h);
const char* mmdm_handle_error(mmdm_handle h);

/* Copy one parameter (reference state_dict key of `Mixer`, e.g. "denoiser1.blocks.0.sa_block.attention.in_proj_weight",
 * "influence.out.weight", "motion_embed.weight", "embed_timestep.time_embed.0.bias"; src/models/mixermdm.py:134-148) from
 * device memory `src` ([rows, cols] row-major; cols = 1 for vectors) into the handle's packed layout.
 * "*.sequence_pos_encoder.pe" buffers are accepted and ignored (tables are regenerated: utils.py:24-35). */
int mmdm_set_weight(mmdm_handle h, const char* name, const float* src, int64_t rows, int64_t cols, void* stream);

/* stats: HOST pointer, 4*262 floats [mean_hml | std_hml | mean_ih | std_ih]  (src/utils/utils.py:44-82). */
int mmdm_set_norm_stats(mmdm_handle h, const float* stats_host);

/* Schedule (HOST pointers): S respaced steps; timestep_map[S] (original t the model sees, gaussian_diffusion.py:2200-2205);
 * coef [4*S] fp32 as in mmdm_xstart_ddim_f32.  Builds the per-step timestep-embedding tables. */
int mmdm_set_schedule(mmdm_handle h, const int* timestep_map, const float* coef, int S, void* stream);

/* Dual sampler only (single_only == 3): w[S] (HOST floats) = the composition weight s_composition(timestep_map[i]) of every respaced
 * step (ClassifierFreeSampleDualMDM.weight, cfg_sampler.py:113-125).  Call after every mmdm_set_schedule. */
int mmdm_set_dual_weights(mmdm_handle h, const float* w_host, int S);

/* Check that every weight is present; allocate nothing afterwards. */
int mmdm_prepare(mmdm_handle h);

/* Begin a sampling call: cond [B, 8*text_dim] (layout src/models/mixermdm.py:342-354) or [B, text_dim] (single_only),
 * x_T [B,T,524] (or [B,T,262]); both chains start from x_T (gaussian_diffusion.py:1863).  Precomputes the text embeddings. */
int mmdm_begin(mmdm_handle h, const float* cond, const float* x_T, int B, int T, void* stream);

/* RAGGED sampling call: B items of DIFFERENT lengths in one batch -- the shape of the reference's evaluation callers, which sample one item at
 * a time with that item's own length (src/evaluation/datasets.py:58, 100-116: B = 1 or mm_num_repeats per call; per-item `motion_lens`), and
 * of any service that batches requests.  lens_host: B HOST ints (frames of every item, 1 .. max_frames; consumed before the call returns);
 * x_T: the items' frames back to back, [sum(lens), 524] (or 262); cond as mmdm_begin.  Layout during the call: the B items form a GROUP of
 * `rows` frame rows (sum(lens) rounded up to the handle's row bucket, MMDM_RAG_BUCKET, default 128; padding rows are zero-initialised,
 * processed like any row and never read by a real one); every buffer of k B sequences is k groups.  Where a sequence starts, its length and
 * the (sequence, frame) of every row are DEVICE arrays written on `stream` by this call: GEMMs and row kernels see sum(lens) rows, the
 * attention / PE / AdaLN-conditioning / geometry kernels index through the maps, and a captured step graph depends on (B, rows, query tiles of the
 * longest item, S) only -- it is reused by every ragged batch of the same bucket.  Every item's result is BIT-IDENTICAL to sampling it alone
 * with mmdm_begin (no kernel's arithmetic depends on a row's position in the batch; tests/test_gpu_ragged.py).  mmdm_get_state then points at
 * [rows, 524] buffers whose first sum(lens) rows are the items back to back; history slots (mmdm_set_history) are [2 * rows, C] with the
 * uncond half at row `rows`; mmdm_call_rows returns (rows, sum(lens)).  Covers the two-chain sampler and the single-person sampler over
 * in2IN / InterGen denoisers with head sizes 64 / 128, every precision mode; B <= min(max_batch, 256), sum(lens) <= max_batch * max_frames;
 * otherwise MMDM_ERR_UNSUPPORTED / MMDM_ERR_ARG.  mmdm_run / mmdm_seek / mmdm_set_history as after mmdm_begin. */
int mmdm_begin_ragged(mmdm_handle h, const float* cond, const float* x_T, int B, const int* lens_host, void* stream);
/* Frame rows per half of the CFG-doubled batch in the begun call's buffers (uniform: B * T), the frames that are real (ragged: sum(lens)),
 * and whether the call is ragged.  Each pointer may be NULL. */
int mmdm_call_rows(mmdm_handle h, int* rows, int* real_rows, int* ragged);

/* Optional history side outputs (src/models/mixermdm.py:794-796, 805-808), CFG-doubled batch 2B.  Each pointer may be NULL.
 * Slot k of a buffer receives the step whose position in the loop is k*every (k = 0 .. ceil(S/every)-1).
 * influence_i1/i2: [slots, 2B, T, 262] for mixing modes 3-4 and [slots, 2B, T, 1] for modes 1-2 (the reference appends the tensor as
 * it stands before the blend: expanded per channel in modes 3-4, one value per frame in modes 1-2, mixermdm.py:739-745);
 * out1/out2/out_influenced: [slots, 2B, T, 524].  Call after mmdm_begin (which resets the call to "no history"); the destinations are
 * written to a device-side descriptor on mmdm_begin's stream, so captured step graphs do not depend on them. */
int mmdm_set_history(mmdm_handle h, float* influence_i1, float* influence_i2, float* out1, float* out2, float* out_influenced, int every);

/* Run `nsteps` consecutive DDIM steps starting at the handle's current position (S-1 after mmdm_begin, counting down).
 * use_graph != 0: one step is captured into a hipGraph on first use and replayed; captured graphs are kept in a least-recently-used
 * cache keyed by (B, T, S) (8 entries; MMDM_GRAPH_CACHE=n overrides), so a caller that alternates shapes -- the evaluation loops of
 * src/evaluation/datasets.py:101-122, 438 call the sampler per item with per-sample T -- re-captures nothing.
 * = MixerDiffusion.ddim_sample_loop_progressive body  gaussian_diffusion.py:1871-1899. */
int mmdm_run(mmdm_handle h, int nsteps, int use_graph, void* stream);
/* Counters of the graph cache: steps captured so far, steps replayed, entries currently cached (each pointer may be NULL). */
int mmdm_graph_stats(mmdm_handle h, int64_t* captures, int64_t* replays, int* cached);
/* Graph execs of the PROCESS that outlived their cache entry (kept, not destroyed: see mmdm_create_shared).  With the no-eviction rule these are
 * the execs of destroyed handles only; a process that never lets two handles share weights reads 0. */
int mmdm_graph_parked(void);

/* Move a begun call to respaced step `step_index` (S-1 = first step of the loop, 0 = last): the next mmdm_run continues from there
 * with the chains as they stand (history slots follow the loop position S-1-step_index).  Lets a caller resume a loop, or drive
 * teacher-forced steps -- overwrite x / x2 through mmdm_get_state, seek, run one step (the parity tests' ddim1000 first/last-20). */
int mmdm_seek(mmdm_handle h, int step_index, void* stream);

/* Device pointers owned by the handle, valid until destroy: current chains and the last pred_xstart(2). */
int mmdm_get_state(mmdm_handle h, float** x, float** x2, float** pred_xstart, float** pred_xstart2, float** model_out);

/* The sampling call's result so far -- the last pred_xstart2 (two-chain sampler; MixerDiffusion.ddim_sample_loop's return value,
 * gaussian_diffusion.py:1820) or pred_xstart (single-chain samplers) -- copied into caller memory ON `stream` (no host synchronisation):
 * [B, T, 524 or 262], ragged call: the items back to back [sum(lens), .].  With mmdm_begin / mmdm_set_history / mmdm_run this is a whole
 * sampling call in plain C calls, e.g. from one host thread per shared handle. */
int mmdm_copy_result(mmdm_handle h, float* dst, void* stream);

/* Teacher-forced pieces for parity tests (operate on caller buffers, CFG-doubled batch n = 2B rows in x/cond):
 *   which: 0 = denoiser1 (individual; x [n,T,262], cond [n,text_dim]) -> out [n,T,262]
 *          1 = denoiser2 (interaction; x [n,T,524], cond [n,3*text_dim]) -> out [n,T,524]
 *          2 = Mixer.forward (x = x1 [n,T,524], x2 [n,T,524], cond [n,8*text_dim]) -> out [n,T,524] (out_influenced)
 *          3 = denoiser1 in "dual_individual" mode (dual handle; x [n,T,524], cond [n,5*text_dim]) -> out [n,T,524]
 *          4 = ClassifierFreeSampleModelX2.forward (src/models/utils/cfg_sampler.py:38-56): n = B UN-doubled rows of x, x2 [B,T,524] and
 *              cond [B,8*text_dim] -> out [B,T,524] = s*Mixer(cond rows) + (1-s)*Mixer(zero-cond rows), s = cfg_scale
 *          (which = 0 with model1_kind = 1: MDMDenoiser.forward, cond [n, d1_latent])
 * t = original (remapped) timestep shared by all rows.  Replaces in2INDenoiser.forward / Mixer.forward / the CFG wrapper's forward.
 * A schedule set with mmdm_set_schedule survives the call; a sampling call in progress does not (call mmdm_begin again). */
int mmdm_module_forward(mmdm_handle h, int which, const float* x, const float* x2, const float* cond, int t,
                        float* out, int n, int T, void* stream);

/* Live kernel timing for bench.py: wraps hipEvents on `stream` around every launch of the kernel class `which`
 * (0 = GEMMs in the handle's own operand type: fp32 / bf16 / split planes, 1 = attention, 2 = fp8-operand GEMMs of precision 3, 3 = the fp32
 * GEMMs of a low-precision handle: embeddings, conditioning, heads -- each class priced against its own matrix peak) during
 * mmdm_run(use_graph=0) and accumulates. */
int mmdm_profile_enable(mmdm_handle h, int on);
int mmdm_profile_read(mmdm_handle h, int which, double* total_ms, int64_t* launches, double* flops, double* algorithmic_bytes);

/* ------------------------------------------------------------------------------------------------
 * 4. Diagnostics (tools/ and bench.py's in-loop clock measurement; not used by any product path).
 * ---------------------------------------------------------------------------------------------- */
/* Sets one PROCESS-GLOBAL diagnostic switch; not thread-safe, never needed to use the library.  Timing ablations and in-kernel stamps live
 * in separate DIAGNOSTIC kernel instantiations that are only launched while a switch asks for them: the kernels a handle launches by
 * default contain no diagnostic code.  Keys (value -1 / 0 = back to normal):
 *   "bf16_tst"                                   0 = the bf16 / fp8 GEMMs store their results directly (row-per-lane) instead of through the
 *                                                workgroup's LDS transposition (bit-identical results; A/B timing)
 *   "gemm_cfg" / "split_cfg" / "bf16_cfg"       force a tile configuration of the fp32 / fp32-split / bf16-fp8 GEMM dispatch (-1 = automatic)
 *   "gemm_tail"                                  force the row-split rule of the fp32 dispatch (t/10 of a round; -1 = the caller's handle decides)
 *   "gemm_s16"                                   small-launch forms of the fp32 dispatch (gemm_s16_kernel / gemm_mix_kernel: 16 x 16-block chains, bit-identical
 *                                                results): -1 = automatic, 0 = off (the 64 x 64-tile launch), 13 / 14 / 23 / 24 = force <blocks per wave, stages>
 *   "fp8p"                                       the persistent fp8 GEMM (gemm_fp8p_kernel, bit-identical): 0 = off (default), 1 = on the cross-attention
 *                                                projections, 2 = wherever it covers the call
 *   "attn_kc32"                                  0 = the 16-key form of the all-bf16 attention instead of the 32-key one (A/B; results differ in rounding)
 *   "gemm_ablate" / "split_ablate" / "attn_ablate"   timing-ablation bits (wrong results)
 *   "gemm_stamps" / "attn_stamps" / "split_timeline" / "bf16_timeline"   device pointer (as an integer) of a stamp buffer, 0 = off
 * Returns MMDM_ERR_ARG for an unknown key. */
int mmdm_diag_set(const char* key, long long value);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MMDM_H */
