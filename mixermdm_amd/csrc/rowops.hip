// Row-wise HBM-bound kernels of the denoising path (gfx950): AdaLN apply, conditioning-vector SiLU,
// time mean, Influence head.  One 64-lane wavefront per row, 16-byte loads, wave-shuffle reductions.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

// This file is built TWICE (mixermdm_amd/build.py): rowops.o as it stands, and rowops_nopk.o with -DMMDM_ROWOPS_NOPK and without the packed-fp32
// VALU instructions -- the same kernels (inside `nopk::`, so that a kernel trace tells them apart) behind entry points named *_nopk, which the
// handles of precision 1-3 take (mmdm.hip ROWOP): their row kernels run beside packed-W GEMMs, where dense packed-fp32 code was seen to
// compute other bits on gfx950 (build.py has the story; AdaLN itself was never seen to move -- tests/test_gpu_hazard.py holds that -- the
// second build removes the question instead of answering it).  The stateless entry points of include/mmdm.h exist in the first build only.
#ifdef MMDM_ROWOPS_NOPK
#define RO(name) name##_nopk
#else
#define RO(name) name
#endif

namespace {
#ifdef MMDM_ROWOPS_NOPK
inline namespace nopk {
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef MMDM_ROWOPS_NOPK
// All-reduce over the wave WITHOUT the LDS (the second build only: the first keeps the bits of rounds 1-5).  __shfl_xor is a ds_bpermute: an LDS round
// trip per butterfly step, six in sequence per reduction, and AdaLN makes two (three with the fp8 row maximum) per row -- the fp8 form ran at 4.45 TB/s
// against the 5.7 of the two-reduction fp32 form although it moves fewer bytes.  Here: four DPP steps inside a 16-lane row (quad_perm 1032, quad_perm
// 2301, row_half_mirror, row_mirror: a VALU modifier, no LDS) and gfx950's v_permlane16_swap / v_permlane32_swap across the four rows (attn_f32.hip).
template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = a + b; b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp<0xB1>(v)); v = fmaxf(v, dpp<0x4E>(v)); v = fmaxf(v, dpp<0x141>(v)); v = fmaxf(v, dpp<0x140>(v));
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b); b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
#else
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
#endif

// out = LN(h) * (1 + scale) + shift ; one wave per row; row cached in registers (D <= 64*4*MAXV).
// Reference: AdaLN.forward src/models/utils/layers.py:15-25 (LayerNorm eps 1e-6, biased variance, no affine).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_fp8x4(f32x4 v) {      // OCP e4m3, RNE, saturating at +-448 (as in gemm_bf16.hip)
    float c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = fminf(fmaxf(v[i], -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
    return (unsigned)w;
}

// OMODE: 0 fp32, 1 bf16, 2 the two fp16 planes of the fp32-split mode (kernels.h mmdm_split2, plane stride rows*D), 3 fp8 e4m3 with a per-row scale (row_scale[row] =
// max|y| / 448; the fp8 GEMM multiplies it back in its epilogue)
template <int MAXV, int OMODE>
__global__ __launch_bounds__(256, MAXV <= 4 ? 8 : 4) void adaln_kernel(const float* __restrict__ h, const float* __restrict__ ss, int ss_ld, int ss_rows,
                                                     void* __restrict__ outv, int rows, int T, int D, float* __restrict__ row_scale = nullptr,
                                                     const int* __restrict__ row_seq = nullptr) {     // ragged batch: sequence of every row (T unused)
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;                                   // float4 per row (D % 4 == 0 checked on host)
    // persistent: a fixed grid of wave slots walks the rows (one launch of ~2000 blocks instead of rows/4 short-lived ones)
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const float* hp = h + (size_t)row * D;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) v[i] = *reinterpret_cast<const f32x4*>(hp + 4 * c);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 d = v[i] - mean;
            q += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-6f);
    const int sq = row_seq ? row_seq[row] : row / T;              // (wave-uniform)
    const float* sp = ss + (size_t)(sq % ss_rows) * ss_ld;
    if constexpr (OMODE == 3) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(sp + 4 * c);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(sp + D + 4 * c);
                v[i] = (v[i] - mean) * rstd * (1.0f + sc) + sh;
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3]))));
            }
        }
        amax = wave_max(amax);
        const float scl = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f, inv = 1.0f / scl;
        if (lane == 0) row_scale[row] = scl;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) *reinterpret_cast<unsigned*>(static_cast<unsigned char*>(outv) + (size_t)row * D + 4 * c) = pack_fp8x4(v[i] * inv);
        }
        continue;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sp + 4 * c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(sp + D + 4 * c);
            const f32x4 y = (v[i] - mean) * rstd * (1.0f + sc) + sh;
            if (OMODE == 2) {
                mmdm_h4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const _Float16 t = mmdm_split_hi(y[e]); oh[e] = t; ol[e] = mmdm_split_lo(y[e], t); }
                _Float16* op = static_cast<_Float16*>(outv) + (size_t)row * D + 4 * c;
                const size_t plane = (size_t)rows * D;
                *reinterpret_cast<mmdm_h4*>(op) = oh;
                *reinterpret_cast<mmdm_h4*>(op + plane) = ol;
            } else if (OMODE == 1) {
                const bf16x4 o = {(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
                *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(outv) + (size_t)row * D + 4 * c) = o;
            } else {
                *reinterpret_cast<f32x4*>(static_cast<float*>(outv) + (size_t)row * D + 4 * c) = y;
            }
        }
    }
    }
}

// out = LN_{eps}(x) * gamma + beta (nn.LayerNorm with affine); one wave per row, row cached in registers, in place allowed.
// Reference: norm1 / norm2 of nn.TransformerEncoderLayer (src/models/mdm.py:252-264, src/models/mixermdm.py:246-259), clip_ln, ln_final.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ out, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int nv = D >> 2;
    // persistent like adaln_kernel: the host caps the grid at 2048 blocks, so every wave slot walks rows with the grid's stride
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const float* xp = x + (size_t)row * D;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) v[i] = *reinterpret_cast<const f32x4*>(xp + 4 * c);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 d = v[i] - mean;
            q += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
            *reinterpret_cast<f32x4*>(out + (size_t)row * D + 4 * c) = (v[i] - mean) * rstd * g + b;
        }
    }
    }
}

// MDMDenoiser sequence assembly (src/models/mdm.py:279-294): dst [nseq, T+1, D];
//   dst[s, 0, :]   = (cond[s, :] + time_tab[*step, :]) + pe[0, :]         (the conditioning token; cond row stride ldc)
//   dst[s, 1+t, :] = src[s, t, :]                                          (pose embeddings, already + pe[1+t])
__global__ void mdm_pack_kernel(const float* __restrict__ src, const float* __restrict__ cond, int ldc, const float* __restrict__ time_tab,
                                const int* __restrict__ step_idx, const float* __restrict__ pe, float* __restrict__ dst, int nseq, int T, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nseq * (T + 1) * D) return;
    const int d = (int)(i % D);
    const size_t row = i / D;
    const int t = (int)(row % (T + 1));
    const size_t s = row / (T + 1);
    dst[i] = t == 0 ? (cond[s * ldc + d] + time_tab[(size_t)(*step_idx) * D + d]) + pe[d] : src[(s * T + (t - 1)) * D + d];
}

// dst[s, t, :] = src[s, 1+t, :]   (drop the conditioning token: mdm.py:296)
__global__ void mdm_unpack_kernel(const float* __restrict__ src, float* __restrict__ dst, int nseq, int T, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nseq * T * D) return;
    const int d = (int)(i % D);
    const size_t row = i / D;
    dst[i] = src[(row + row / T + 1) * D + d];
}

// out[b, l, :] = table[tokens[b, l], :] + pos[l, :]   (CLIP token + positional embedding: src/models/mixermdm.py:298-299)
__global__ void token_embed_kernel(const float* __restrict__ table, const int* __restrict__ tokens, const float* __restrict__ pos,
                                   float* __restrict__ out, int n, int L, int D, int vocab) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * L * D) return;
    const int d = (int)(i % D);
    const size_t row = i / D;
    int tok = tokens[row];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    out[i] = table[(size_t)tok * D + d] + pos[(row % L) * D + d];
}

__global__ void cond_silu_kernel(const float* __restrict__ time_tab, const int* __restrict__ step_idx, const float* __restrict__ txt,
                                 float* __restrict__ out, int rows, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * D) return;
    const int d = (int)(i % D);
    const float e = time_tab[(size_t)(*step_idx) * D + d] + txt[i];
    out[i] = e / (1.0f + expf(-e));
}

// the same rows as the two fp16 operand planes of the fp32-split GEMM (out[i], out[plane + i]): cond_vectors of the low-precision handles
__global__ void cond_silu_planes_kernel(const float* __restrict__ time_tab, const int* __restrict__ step_idx, const float* __restrict__ txt,
                                        _Float16* __restrict__ out, size_t plane, int rows, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * D) return;
    const int d = (int)(i % D);
    const float e = time_tab[(size_t)(*step_idx) * D + d] + txt[i];
    _Float16 h, l;
    mmdm_split2(e / (1.0f + expf(-e)), h, l);
    out[i] = h;
    out[plane + i] = l;
}

__global__ void mean_time_kernel(const float* __restrict__ h, float* __restrict__ out, int T, int D) {
    const int seq = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float* p = h + (size_t)seq * T * D + d;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += p[(size_t)t * D];
    out[(size_t)seq * D + d] = s / (float)T;
}

// ragged batch: sequence s = rows [seq_off[s], seq_off[s] + seq_len[s]) of h (same summation order as mean_time_kernel: bit-identical per sequence)
__global__ void mean_time_rag_kernel(const float* __restrict__ h, float* __restrict__ out, const int* __restrict__ seq_off, const int* __restrict__ seq_len, int D) {
    const int seq = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int T = seq_len[seq];
    const float* p = h + (size_t)seq_off[seq] * D + d;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += p[(size_t)t * D];
    out[(size_t)seq * D + d] = s / (float)T;
}

// w[row, o] = sigmoid(h[row,:] . Wout[o,:] + b[o]); one wave per row, nw <= 23 outputs.
// Reference: Influence.forward tail src/models/utils/influence.py:124-125.
__global__ __launch_bounds__(256) void influence_head_kernel(const float* __restrict__ h, const float* __restrict__ W, const float* __restrict__ b,
                                                              float* __restrict__ w, int rows, int D, int nw) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* hp = h + (size_t)row * D;
    for (int o = 0; o < nw; ++o) {
        const float* wp = W + (size_t)o * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += hp[c] * wp[c];
        s = wave_sum(s);
        if (lane == 0) {
            const float z = s + b[o];
            w[(size_t)row * nw + o] = 1.0f / (1.0f + expf(-z));
        }
    }
}

// scipy.ndimage.gaussian_filter1d(x, sigma, axis=time, mode="nearest") on [n, T, C]: correlate with a symmetric kernel of
// radius r, indices clamped to [0, T-1]; weights and accumulation in double (scipy's correlate1d accumulates in double).
__global__ void gauss1d_kernel(const float* __restrict__ x, float* __restrict__ out, const double* __restrict__ w, int radius, int n, int T, int C) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * T * C) return;
    const int c = (int)(idx % C);
    const int t = (int)((idx / C) % T);
    const size_t b = idx / ((size_t)C * T);
    const float* xb = x + b * T * C + c;
    double acc = 0.0;
    for (int k = -radius; k <= radius; ++k) {
        int tt = t + k;
        tt = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        acc += w[k + radius] * (double)xb[(size_t)tt * C];
    }
    out[idx] = (float)acc;
}

#ifdef MMDM_ROWOPS_NOPK
}  // namespace nopk
#endif
}  // namespace

#ifndef MMDM_ROWOPS_NOPK
extern "C" int mmdm_gaussian_filter1d_f32(const float* x, float* out, const double* weights, int radius, int n, int T, int C, void* stream) {
    if (n == 0 || T == 0 || C == 0) return MMDM_OK;
    if (!x || !out || !weights || radius < 0 || n < 0 || T < 0 || C < 0 || x == out)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_gaussian_filter1d_f32: bad arguments (in-place is not supported)");
    const size_t total = (size_t)n * T * C;
    hipLaunchKernelGGL(gauss1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, weights, radius, n, T, C);
    return mmdm_check_launch("gauss1d");
}

extern "C" int mmdm_adaln_f32(const float* h, const float* ss, int ss_ld, int ss_rows, float* out, int nseq, int T, int D, void* stream) {
    return mmdm_adaln_ex(h, ss, ss_ld, ss_rows, out, 0, nseq, T, D, stream);
}

extern "C" int mmdm_adaln_ex(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, int out_bf16, int nseq, int T, int D, void* stream) {
    if (out_bf16 < 0 || out_bf16 > 2) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_adaln_ex: output mode must be 0 (fp32), 1 (bf16) or 2 (the two fp16 planes of the fp32-split mode)");
    return mmdm_adaln_any(h, ss, ss_ld, ss_rows, out, out_bf16, nullptr, nseq, T, D, stream);
}

extern "C" int mmdm_adaln_fp8(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, float* row_scale, int nseq, int T, int D, void* stream) {
    if (!row_scale) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_adaln_fp8: null row_scale");
    return mmdm_adaln_any(h, ss, ss_ld, ss_rows, out, 3, row_scale, nseq, T, D, stream);
}

#endif  // !MMDM_ROWOPS_NOPK

// row_seq != nullptr: ragged batch -- `rows_rag` rows in all, row r belongs to sequence row_seq[r] (device array); nseq / T are then unused
int RO(mmdm_adaln_any)(const float* h, const float* ss, int ss_ld, int ss_rows, void* out, int out_bf16, float* row_scale, int nseq, int T, int D, void* stream,
                   const int* row_seq, int rows_rag) {
    if (row_seq) { nseq = 1; T = rows_rag; }
    if (nseq == 0 || T == 0) return MMDM_OK;
    if (!h || !ss || !out || nseq < 0 || T < 0 || D <= 0 || ss_rows <= 0 || ss_ld < 2 * D)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_adaln_f32: bad arguments nseq=%d T=%d D=%d ss_ld=%d ss_rows=%d", nseq, T, D, ss_ld, ss_rows);
    if ((D & 3) || (ss_ld & 3) || ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(ss) | reinterpret_cast<uintptr_t>(out)) & 15))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_adaln_f32: D and ss_ld must be multiples of 4 and pointers 16-byte aligned");
    if (D > 64 * 4 * 8) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_adaln_f32: D=%d > 2048", D);
    const int rows = nseq * T;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(min((rows + 3) / 4, 2048)), block(256);          // 256 CUs x 8 resident blocks
#define ADALN_LAUNCH(V)                                                                                                   \
    do {                                                                                                                  \
        if (out_bf16 == 3) hipLaunchKernelGGL((adaln_kernel<V, 3>), grid, block, 0, st, h, ss, ss_ld, ss_rows, out, rows, T, D, row_scale, row_seq);       \
        else if (out_bf16 == 2) hipLaunchKernelGGL((adaln_kernel<V, 2>), grid, block, 0, st, h, ss, ss_ld, ss_rows, out, rows, T, D, (float*)nullptr, row_seq);       \
        else if (out_bf16) hipLaunchKernelGGL((adaln_kernel<V, 1>), grid, block, 0, st, h, ss, ss_ld, ss_rows, out, rows, T, D, (float*)nullptr, row_seq);       \
        else hipLaunchKernelGGL((adaln_kernel<V, 0>), grid, block, 0, st, h, ss, ss_ld, ss_rows, out, rows, T, D, (float*)nullptr, row_seq);                     \
    } while (0)
    if (D <= 256) ADALN_LAUNCH(1);
    else if (D <= 512) ADALN_LAUNCH(2);
    else if (D <= 1024) ADALN_LAUNCH(4);
    else ADALN_LAUNCH(8);
#undef ADALN_LAUNCH
    return mmdm_check_launch("adaln");
}

extern "C" int RO(mmdm_layernorm_f32)(const float* x, const float* gamma, const float* beta, float* out, int rows, int D, float eps, void* stream) {
    if (rows == 0) return MMDM_OK;
    if (!x || !gamma || !beta || !out || rows < 0 || D <= 0 || !(eps > 0.f)) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_layernorm_f32: bad arguments rows=%d D=%d", rows, D);
    if ((D & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(out)) & 15))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_layernorm_f32: D must be a multiple of 4 and pointers 16-byte aligned");
    if (D > 64 * 4 * 8) return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_layernorm_f32: D=%d > 2048", D);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(min((rows + 3) / 4, 2048)), block(256);          // 256 CUs x 8 resident blocks
    if (D <= 256) hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, st, x, gamma, beta, out, rows, D, eps);
    else if (D <= 512) hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, st, x, gamma, beta, out, rows, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, st, x, gamma, beta, out, rows, D, eps);
    else hipLaunchKernelGGL((layernorm_kernel<8>), grid, block, 0, st, x, gamma, beta, out, rows, D, eps);
    return mmdm_check_launch("layernorm");
}

int RO(mmdm_mdm_pack)(const float* src, const float* cond, int ldc, const float* time_tab, const int* step_idx, const float* pe, float* dst,
                  int nseq, int T, int D, hipStream_t st) {
    const size_t n = (size_t)nseq * (T + 1) * D;
    if (n == 0) return MMDM_OK;
    hipLaunchKernelGGL(mdm_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, cond, ldc, time_tab, step_idx, pe, dst, nseq, T, D);
    return mmdm_check_launch("mdm_pack");
}

int RO(mmdm_mdm_unpack)(const float* src, float* dst, int nseq, int T, int D, hipStream_t st) {
    const size_t n = (size_t)nseq * T * D;
    if (n == 0) return MMDM_OK;
    hipLaunchKernelGGL(mdm_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, nseq, T, D);
    return mmdm_check_launch("mdm_unpack");
}

#ifndef MMDM_ROWOPS_NOPK
extern "C" int mmdm_token_embed_f32(const float* table, int vocab, const int* tokens, const float* pos, float* out, int n, int L, int D, void* stream) {
    if (n == 0 || L == 0) return MMDM_OK;
    if (!table || !tokens || !pos || !out || n < 0 || L < 0 || D <= 0 || vocab <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_token_embed_f32: bad arguments");
    const size_t total = (size_t)n * L * D;
    hipLaunchKernelGGL(token_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), table, tokens, pos, out, n, L, D, vocab);
    return mmdm_check_launch("token_embed");
}

#endif

extern "C" int RO(mmdm_cond_silu_f32)(const float* time_tab, const int* step_idx, const float* txt, float* out, int rows, int D, void* stream) {
    if (rows == 0) return MMDM_OK;
    if (!time_tab || !step_idx || !txt || !out || rows < 0 || D <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_cond_silu_f32: bad arguments");
    const size_t n = (size_t)rows * D;
    hipLaunchKernelGGL(cond_silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), time_tab, step_idx, txt, out, rows, D);
    return mmdm_check_launch("cond_silu");
}

int RO(mmdm_cond_silu_planes)(const float* time_tab, const int* step_idx, const float* txt, _Float16* out, size_t plane, int rows, int D, hipStream_t st) {
    if (rows == 0) return MMDM_OK;
    const size_t n = (size_t)rows * D;
    hipLaunchKernelGGL(cond_silu_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, time_tab, step_idx, txt, out, plane, rows, D);
    return mmdm_check_launch("cond_silu_planes");
}

extern "C" int RO(mmdm_mean_time_f32)(const float* h, float* out, int nseq, int T, int D, void* stream) {
    if (nseq == 0) return MMDM_OK;
    if (!h || !out || nseq < 0 || T <= 0 || D <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_mean_time_f32: bad arguments");
    hipLaunchKernelGGL(mean_time_kernel, dim3((D + 255) / 256, nseq), dim3(256), 0, static_cast<hipStream_t>(stream), h, out, T, D);
    return mmdm_check_launch("mean_time");
}

int RO(mmdm_mean_time_rag)(const float* h, float* out, int nseq, const int* seq_off, const int* seq_len, int D, hipStream_t st) {
    if (nseq == 0) return MMDM_OK;
    hipLaunchKernelGGL(mean_time_rag_kernel, dim3((D + 255) / 256, nseq), dim3(256), 0, st, h, out, seq_off, seq_len, D);
    return mmdm_check_launch("mean_time_rag");
}

#ifndef MMDM_ROWOPS_NOPK
extern "C" int mmdm_influence_head_f32(const float* h, const float* Wout, const float* bout, float* w, int rows, int D, int nw, void* stream) {
    if (rows == 0) return MMDM_OK;
    if (!h || !Wout || !bout || !w || rows < 0 || D <= 0 || nw <= 0 || nw > 23)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_influence_head_f32: bad arguments rows=%d D=%d nw=%d", rows, D, nw);
    hipLaunchKernelGGL(influence_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), h, Wout, bout, w, rows, D, nw);
    return mmdm_check_launch("influence_head");
}
#endif
