// Exact-fp32 multi-head attention core with add_zero_attn for gfx950 (v_mfma_f32_16x16x4_f32, flash-style).
//
// Replaces the scaled-dot-product attention inside nn.MultiheadAttention(add_zero_attn=True) as used by
// VanillaSelfAttention / VanillaCrossAttention (reference: src/models/utils/layers.py:33-44, 74-87).
//   out[s,q,h,:] = softmax_{keys + one zero key}( Q[s,q,h,:].K[s',k,h,:] / sqrt(dh) ) V[s',k,h,:],  s' = (s+shift) % nseq
// The extra key has logit 0 and value 0, so it is folded in as the INITIAL online-softmax state (m=0, l=1, O=0).
//
// Structure: one 256-thread workgroup = 4 waves = 64 queries of one (sequence, head); each wave owns 16 queries.
// Keys/values stream through LDS in chunks of 64.  QK^T is computed swapped (S^T = K Q^T) so that each lane holds
// four keys of ONE query column (C/D map: col = lane&15 = query, row = 4*(lane>>4)+reg = key): the row max / row sum
// are 4 local values + two xor-shuffles (16, 32), and the exponentiated tile is already the A operand of the
// P.V MFMA (k index = lane group) with no cross-lane movement or LDS round trip.  The d (reduction) order of
// QK^T is permuted identically for both operands so that one 16-byte LDS read feeds four MFMAs.
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 64;   // keys per LDS chunk
constexpr int QW = 16;   // queries per wave
constexpr int QB = 64;   // queries per workgroup

struct AttnArgs {
    const float* Q; const float* K; const float* V; float* O;
    int ldq, ldk, ldv, ldo;
    int nseq, Tq, Tk, H, shift, qtiles;
    float scale;
};

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(AttnArgs p) {
    constexpr int LDK = DH + 4;                 // padded LDS row (floats): 16-byte reads spread over bank slots
    constexpr int NJ = DH / 16;                 // d groups of 16 (QK^T) == 16-wide output column tiles (PV)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                           // [KC][LDK]
    float* Vs = smem + KC * LDK;                // [KC][LDK]

    const int bid = blockIdx.x;
    const int qt = bid % p.qtiles;
    const int sh = bid / p.qtiles;
    const int head = sh % p.H;
    const int seq = sh / p.H;
    const int kvseq = (seq + p.shift) % p.nseq;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const int q0 = qt * QB + wave * QW;

    // Q fragment (B operand of S^T = K Q^T): lane (q = lq, g) holds Q[q][16j + 4g + s], pre-scaled.
    f32x4 qf[NJ];
    {
        int qrow = q0 + lq;
        if (qrow >= p.Tq) qrow = p.Tq - 1;
        const float* qp = p.Q + ((size_t)seq * p.Tq + qrow) * p.ldq + head * DH + 4 * g;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + 16 * j);
            qf[j] = v * p.scale;
        }
    }

    f32x4 o[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = 0.f, l_run = 1.f;             // add_zero_attn: one key with logit 0, value 0 already absorbed

    const float* Kg = p.K + (size_t)kvseq * p.Tk * p.ldk + head * DH;
    const float* Vg = p.V + (size_t)kvseq * p.Tk * p.ldv + head * DH;

    for (int c0 = 0; c0 < p.Tk; c0 += KC) {
        __syncthreads();                        // previous chunk fully consumed
        // stage K and V chunk: KC rows x DH floats each; zero-fill rows past Tk
        constexpr int F4_PER_ROW = DH / 4;
        constexpr int F4_TOTAL = KC * F4_PER_ROW;
#pragma unroll
        for (int i = tid; i < F4_TOTAL; i += 256) {
            const int row = i / F4_PER_ROW, c4 = (i % F4_PER_ROW) * 4;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (c0 + row < p.Tk) {
                kv = *reinterpret_cast<const f32x4*>(Kg + (size_t)(c0 + row) * p.ldk + c4);
                vv = *reinterpret_cast<const f32x4*>(Vg + (size_t)(c0 + row) * p.ldv + c4);
            }
            *reinterpret_cast<f32x4*>(&Ks[row * LDK + c4]) = kv;
            *reinterpret_cast<f32x4*>(&Vs[row * LDK + c4]) = vv;
        }
        __syncthreads();

        // S^T tiles: st[kt][reg] = score(key = c0 + 16kt + 4g + reg, query = lq)
        f32x4 st[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 kf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                kf[kt] = *reinterpret_cast<const f32x4*>(&Ks[(16 * kt + lq) * LDK + 16 * j + 4 * g]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt][s], qf[j][s], st[kt], 0, 0, 0);
        }

        // mask keys past Tk, chunk max for this lane's query
        float cmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = c0 + 16 * kt + 4 * g + r;
                if (key >= p.Tk) st[kt][r] = -INFINITY;
                cmax = fmaxf(cmax, st[kt][r]);
            }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = expf(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                st[kt][r] = expf(st[kt][r] - m_new);
                lsum += st[kt][r];
            }
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;

        // rescale O: accumulator rows are queries 4g + r, whose alpha lives in lanes with (lane&15) == 4g + r
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[j][r] *= ar[r];

        // O[q][n] += sum_key P[q][key] V[key][n]:  A = P (lane-local: st[kt][r] is P[q=lq][key=16kt+4g+r]),
        // B = V[key = 16kt + 4g + r][n = 16j + lq]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = &Vs[(16 * kt + 4 * g + r) * LDK + lq];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[kt][r], vrow[16 * j], o[j], 0, 0, 0);
            }
    }

    // normalise and store: accumulator element (j, r) is O[query q0 + 4g + r][16j + lq]
    float lr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) lr[r] = __shfl(l_run, 4 * g + r);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 4 * g + r;
        if (qrow >= p.Tq) continue;
        const float inv = 1.0f / lr[r];
        float* op = p.O + ((size_t)seq * p.Tq + qrow) * p.ldo + head * DH + lq;
#pragma unroll
        for (int j = 0; j < NJ; ++j) op[16 * j] = o[j][r] * inv;
    }
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
    float m = 0.f, l = 1.f;
    for (int k = 0; k < p.Tk; ++k) {
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
    float* op = p.O + ((size_t)seq * p.Tq + q) * p.ldo + head * DH;
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < DH; ++d) op[d] = acc[d] * inv;
}

template <int DH>
constexpr int attn_smem() { return 2 * KC * (DH + 4) * 4; }

template <int DH>
int launch_mfma(const AttnArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((attn_mfma_kernel<DH>), dim3(a.nseq * a.H * a.qtiles), dim3(256), attn_smem<DH>(), st, a);
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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<128>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, attn_smem<128>());
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, attn_smem<64>());
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(attn): %s", hipGetErrorString(e));
    return MMDM_OK;
}

extern "C" int mmdm_attention_f32(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                  int nseq, int Tq, int Tk, int H, int dh, int kv_seq_shift, void* stream) {
    if (nseq == 0 || Tq == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    if (!Q || !K || !V || !O || nseq < 0 || Tq < 0 || Tk <= 0 || H <= 0 || dh <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: bad shape nseq=%d Tq=%d Tk=%d H=%d dh=%d", nseq, Tq, Tk, H, dh);
    if (ldq < H * dh || ldk < H * dh || ldv < H * dh || ldo < H * dh)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: row strides must cover H*dh=%d", H * dh);
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.nseq = nseq; a.Tq = Tq; a.Tk = Tk; a.H = H;
    a.shift = ((kv_seq_shift % nseq) + nseq) % nseq;
    a.qtiles = (Tq + QB - 1) / QB;
    a.scale = 1.0f / sqrtf((float)dh);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dh == 128 || dh == 64) {
        const bool al = ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V)) & 15) == 0 &&
                        ((ldq | ldk | ldv) & 3) == 0;
        if (!al) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_attention_f32: Q/K/V must be 16-byte aligned with row strides %% 4 == 0");
        return dh == 128 ? launch_mfma<128>(a, st) : launch_mfma<64>(a, st);
    }
    switch (dh) {
        case 4: return launch_small<4>(a, st);
        case 8: return launch_small<8>(a, st);
        case 16: return launch_small<16>(a, st);
        case 32: return launch_small<32>(a, st);
        default: return mmdm_set_error(MMDM_ERR_UNSUPPORTED, "mmdm_attention_f32: head dim %d not supported (4,8,16,32,64,128)", dh);
    }
}
