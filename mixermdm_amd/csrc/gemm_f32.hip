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
#include "kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, LDP = 36;
constexpr int TILE_FLOATS = BM * LDP;                 // one operand, one buffer
constexpr int SMEM_BYTES = 4 * TILE_FLOATS * 4;       // A,B x 2 buffers = 73,728 B -> 2 workgroups / CU

struct GemmArgs {
    const float* A; const float* W; const float* bias; float* C; const float* extra;
    int lda, ldw, ldc, ld_extra;
    int M, N, K, Kw, epilogue, period;   // Kw >= K: readable columns of W (zero beyond K)
    int mt, nt;
};

template <int VEC>
__device__ __forceinline__ f32x4 load4(const float* __restrict__ base, int ld, int row, int nrows, int k, int K) {
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

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

template <int AVEC, int WVEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [2][BM*LDP]
    float* Bs = smem + 2 * TILE_FLOATS;    // [2][BN*LDP]

    // XCD-aware remap (bijective for any grid size): blocks b and b+8 share an XCD; give each XCD a contiguous tile range.
    const int nwg = p.mt * p.nt;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int m0 = (swz / p.nt) * BM;
    const int n0 = (swz % p.nt) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // staging map: 128 rows x 8 float4 per tile; thread -> rows (tid>>3) + 32j, float4 column tid&7
    const int sr = tid >> 3, sc = (tid & 7) * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nkt = (p.K + BK - 1) / BK;
    f32x4 ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ra[j] = load4<AVEC>(p.A, p.lda, m0 + sr + 32 * j, p.M, sc, p.K);
        rb[j] = load4<WVEC>(p.W, p.ldw, n0 + sr + 32 * j, p.N, sc, p.Kw);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<f32x4*>(&As[(sr + 32 * j) * LDP + sc]) = ra[j];
        *reinterpret_cast<f32x4*>(&Bs[(sr + 32 * j) * LDP + sc]) = rb[j];
    }
    __syncthreads();

    const int a_off = (wm * 64 + l31) * LDP + 4 * lh;
    const int b_off = (wn * 64 + l31) * LDP + 4 * lh;

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nkt;
        if (more) {
            const int k0 = (kt + 1) * BK + sc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ra[j] = load4<AVEC>(p.A, p.lda, m0 + sr + 32 * j, p.M, k0, p.K);
                rb[j] = load4<WVEC>(p.W, p.ldw, n0 + sr + 32 * j, p.N, k0, p.Kw);
            }
        }
        const float* Ac = As + cur * TILE_FLOATS + a_off;
        const float* Bc = Bs + cur * TILE_FLOATS + b_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ac + g * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ac + 32 * LDP + g * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bc + g * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bc + 32 * LDP + g * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
            }
        }
        if (more) {
            float* An = As + (cur ^ 1) * TILE_FLOATS;
            float* Bn = Bs + (cur ^ 1) * TILE_FLOATS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<f32x4*>(&An[(sr + 32 * j) * LDP + sc]) = ra[j];
                *reinterpret_cast<f32x4*>(&Bn[(sr + 32 * j) * LDP + sc]) = rb[j];
            }
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row >= p.M) continue;
                float v = acc[i][j][e] + bv;
                switch (p.epilogue) {
                    case MMDM_EPI_BIAS_GELU: v = gelu_erf(v); break;
                    case MMDM_EPI_BIAS_RESID: v += p.extra[(size_t)row * p.ld_extra + col]; break;
                    case MMDM_EPI_BIAS_PE: v += p.extra[(size_t)(row % p.period) * p.ld_extra + col]; break;
                    case MMDM_EPI_BIAS_SILU: v = silu(v); break;
                    default: break;
                }
                p.C[(size_t)row * p.ldc + col] = v;
            }
        }
    }
}

template <int AVEC, int WVEC>
int set_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<AVEC, WVEC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return mmdm_set_error(MMDM_ERR_HIP, "hipFuncSetAttribute(gemm): %s", hipGetErrorString(e));
    return MMDM_OK;
}

template <int AVEC, int WVEC>
int launch(const GemmArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((gemm_f32_kernel<AVEC, WVEC>), dim3(a.mt * a.nt), dim3(256), SMEM_BYTES, st, a);
    return mmdm_check_launch("gemm_f32");
}

inline bool vec_ok(const float* p, int ld, int K) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (K & 3) == 0;
}

}  // namespace

int mmdm_gemm_init(void) {
    int rc;
    if ((rc = set_attr<4, 4>())) return rc;
    if ((rc = set_attr<1, 4>())) return rc;
    if ((rc = set_attr<4, 1>())) return rc;
    return set_attr<1, 1>();
}

extern "C" int mmdm_linear_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                               int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    return mmdm_linear_f32_ex(A, lda, W, ldw, K, bias, C, ldc, M, N, K, epilogue, extra, ld_extra, period, stream);
}

// Kw: number of readable columns in each W row (>= K, columns K..Kw-1 must be zero) -- lets a K = 262 weight that the
// handle stores zero-padded to 264 columns be fetched with 16-byte loads.
int mmdm_linear_f32_ex(const float* A, int lda, const float* W, int ldw, int Kw, const float* bias, float* C, int ldc,
                       int M, int N, int K, int epilogue, const float* extra, int ld_extra, int period, void* stream) {
    if (M == 0 || N == 0) return MMDM_OK;
    if (int rc = mmdm_kernels_init()) return rc;
    if (!A || !W || !C || M < 0 || N < 0 || K <= 0 || lda < K || ldw < Kw || Kw < K || ldc < N)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: bad shape M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
    if (epilogue < MMDM_EPI_BIAS || epilogue > MMDM_EPI_BIAS_SILU)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: unknown epilogue %d", epilogue);
    if ((epilogue == MMDM_EPI_BIAS_RESID || epilogue == MMDM_EPI_BIAS_PE) && (!extra || ld_extra < N))
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: epilogue %d needs `extra` with ld >= N", epilogue);
    if (epilogue == MMDM_EPI_BIAS_PE && period <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_linear_f32: PE epilogue needs period > 0");
    GemmArgs a;
    a.A = A; a.W = W; a.bias = bias; a.C = C; a.extra = extra;
    a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ld_extra = ld_extra;
    a.M = M; a.N = N; a.K = K; a.Kw = Kw; a.epilogue = epilogue; a.period = period > 0 ? period : 1;
    a.mt = (M + BM - 1) / BM; a.nt = (N + BN - 1) / BN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool av = vec_ok(A, lda, K), wv = vec_ok(W, ldw, Kw);
    if (av && wv) return launch<4, 4>(a, st);
    if (!av && wv) return launch<1, 4>(a, st);
    if (av && !wv) return launch<4, 1>(a, st);
    return launch<1, 1>(a, st);
}
