// Geometry, blend and DDIM-update kernels of the Mixer step (gfx950).  HBM-bound elementwise work over
// [n, T, 2 persons, 262] motion tensors; one thread per (sample, person, frame, joint) so that the 21 rotation
// round trips (6D -> matrix -> quaternion -> axis-angle -> quaternion -> matrix -> 6D) spread over lanes.
//
// Reference restated here (value-for-value, branch-free versions of the reference's masked-index code):
//   src/utils/alignment.py:11-67 (ih_to_smpl / smpl_to_ih), :69-158 (align_trajectories / align_motions), :161-222 (center_motion)
//   src/utils/rotation_conversions.py:38-120, 449-571;  src/utils/quaternion.py:54-73 (qrot), 386-396 (qbetween)
//   src/utils/utils.py:44-82 (normalisers);  src/models/mixermdm.py:691-801;  src/models/utils/cfg_sampler.py:49-55
//   src/models/utils/gaussian_diffusion.py:2031-2062 (process_xstart), 1936-1965 (two-chain DDIM, eta = 0), 558-562
#include <hip/hip_runtime.h>
#include <math.h>
#include "kernels.h"

namespace {

constexpr int NF = MMDM_NF;          // 262
constexpr int NF2 = 2 * MMDM_NF;     // 524
constexpr int NJ = MMDM_NJ;          // 22
constexpr int ROT0 = 132;            // first rot6d channel
constexpr int FEET0 = 258;
constexpr int R_HIP = 2, L_HIP = 1;  // FACE_JOINT_INDX[:2]  src/utils/paramUtil.py:89

struct V3 { float x, y, z; };
struct Q4 { float w, x, y, z; };

__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// quaternion.py:54-73
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
    const V3 qv{q.x, q.y, q.z};
    const V3 uv = cross3(qv, v);
    const V3 uuv = cross3(qv, uv);
    return V3{v.x + 2.f * (q.w * uv.x + uuv.x), v.y + 2.f * (q.w * uv.y + uuv.y), v.z + 2.f * (q.w * uv.z + uuv.z)};
}

// quaternion.py:386-396 + qnormalize :28-30
__device__ __forceinline__ Q4 qbetween(V3 a, V3 b) {
    const V3 v = cross3(a, b);
    const float w = sqrtf((a.x * a.x + a.y * a.y + a.z * a.z) * (b.x * b.x + b.y * b.y + b.z * b.z)) + (a.x * b.x + a.y * b.y + a.z * b.z) + 1e-8f;
    const float n = sqrtf(w * w + v.x * v.x + v.y * v.y + v.z * v.z);
    return Q4{w / n, v.x / n, v.y / n, v.z / n};
}

__device__ __forceinline__ float sqrt_pos(float x) { return x > 0.f ? sqrtf(x) : 0.f; }                 // rotation_conversions.py:85-95
__device__ __forceinline__ float copysign_ref(float a, float b) { return ((a < 0.f) != (b < 0.f)) ? -a : a; }  // :68-82
__device__ __forceinline__ float half_sinc(float half, float ang) {                                      // :466-475 / :497-507
    return (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.f) : (sinf(half) / ang);
}

// rot6d (interleaved) -> rot6d after the ih_to_smpl -> smpl_to_ih round trip (the two "* -1" cancel exactly).
__device__ __forceinline__ void rot6d_roundtrip(const float d[6], float out[6]) {
    // rotation_6d_to_matrix :511-534 (a1 = d[0,2,4], a2 = d[1,3,5]; F.normalize eps 1e-12)
    float a1x = d[0], a1y = d[2], a1z = d[4], a2x = d[1], a2y = d[3], a2z = d[5];
    float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
    const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    // rows: m0 = b1, m1 = b2, m2 = b3.   matrix_to_quaternion :98-120
    const float m00 = b1x, m01 = b1y, m02 = b1z, m10 = b2x, m11 = b2y, m12 = b2z, m20 = b3x, m21 = b3y, m22 = b3z;
    const float qw = 0.5f * sqrt_pos(1.f + m00 + m11 + m22);
    const float qx = copysign_ref(0.5f * sqrt_pos(1.f + m00 - m11 - m22), m21 - m12);
    const float qy = copysign_ref(0.5f * sqrt_pos(1.f - m00 + m11 - m22), m02 - m20);
    const float qz = copysign_ref(0.5f * sqrt_pos(1.f - m00 - m11 + m22), m10 - m01);
    // quaternion_to_axis_angle :480-508
    const float nrm = sqrtf(qx * qx + qy * qy + qz * qz);
    const float half = atan2f(nrm, qw);
    const float ang = 2.f * half;
    const float s1 = half_sinc(half, ang);
    const float ax = qx / s1, ay = qy / s1, az = qz / s1;     // (ih_to_smpl's *-1 and smpl_to_ih's *-1 cancel)
    // axis_angle_to_quaternion :449-477
    const float ang2 = sqrtf(ax * ax + ay * ay + az * az);
    const float half2 = 0.5f * ang2;
    const float s2 = half_sinc(half2, ang2);
    const float r = cosf(half2), i = ax * s2, j = ay * s2, k = az * s2;
    // quaternion_to_matrix :38-65 (first two rows), matrix_to_rotation_6d :540-571 (interleave)
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    out[0] = 1.f - two_s * (j * j + k * k);   // m00
    out[2] = two_s * (i * j - k * r);         // m01
    out[4] = two_s * (i * k + j * r);         // m02
    out[1] = two_s * (i * j + k * r);         // m10
    out[3] = 1.f - two_s * (i * i + k * k);   // m11
    out[5] = two_s * (j * k - i * r);         // m12
}

// ---------------------------------------------------------------------------------------------------------
// Mixer pre-processing: denormalise both denoiser outputs, align the individual prediction to the interaction one.
// grid: n * 2 persons * T blocks of 32 threads?  -> flat: one thread per (b, p, t, j), j in [0, 22).
// ---------------------------------------------------------------------------------------------------------
// RAG: ragged batch (kernels.h mmdm_rag) -- `n` counts GROUPS of rg.rows frame rows (cond | uncond halves), a thread is (frame row, person, joint),
// frame 0 / the last frame of its item come from the row maps; padding rows leave at once.  The arithmetic is the uniform kernel's.
template <bool RAG>
__global__ __launch_bounds__(256) void mixer_pre_kernel(const float* __restrict__ o1, const float* __restrict__ o2, const float* __restrict__ stats,
                                                         float* __restrict__ out1, float* __restrict__ out2, int n, int T, int align, mmdm_rag rg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int j, t, p;
    size_t seq;
    if constexpr (RAG) {
        if (idx >= n * rg.rows * 2 * NJ) return;
        j = idx % NJ;
        p = (idx / NJ) % 2;
        const int r = idx / (NJ * 2), rl = r % rg.rows, item = rg.row_item[rl];
        if (item < 0) return;
        t = rg.row_pos[rl];
        T = rg.item_len[item];
        seq = (size_t)(r - t) * NF2 + (size_t)p * NF;
    } else {
        const int total = n * 2 * T * NJ;
        if (idx >= total) return;
        j = idx % NJ;
        t = (idx / NJ) % T;
        p = (idx / (NJ * T)) % 2;
        const int b = idx / (NJ * T * 2);
        seq = (size_t)b * T * NF2 + (size_t)p * NF;      // start of (b, frame 0, person p)
    }
    const float* mh = stats, *sh = stats + NF, *mi = stats + 2 * NF, *si = stats + 3 * NF;
    const size_t off = seq + (size_t)t * NF2;
    const float* a = o1 + off;   // individual denoiser, HML3D normalisation
    const float* c = o2 + off;   // interaction denoiser, InterHuman normalisation
    float* y1 = out1 + off;
    float* y2 = out2 + off;

    // interaction stream: pos / vel pass through (denormalised); rot6d re-orthonormalised when align
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        y2[3 * j + k] = c[3 * j + k] * si[3 * j + k] + mi[3 * j + k];
        y2[66 + 3 * j + k] = c[66 + 3 * j + k] * si[66 + 3 * j + k] + mi[66 + 3 * j + k];
    }
    if (j < 21) {
        float d[6], r[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = c[ROT0 + 6 * j + k] * si[ROT0 + 6 * j + k] + mi[ROT0 + 6 * j + k];
        if (align) rot6d_roundtrip(d, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) y2[ROT0 + 6 * j + k] = align ? r[k] : d[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = a[ROT0 + 6 * j + k] * sh[ROT0 + 6 * j + k] + mh[ROT0 + 6 * j + k];
        if (align) rot6d_roundtrip(d, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) y1[ROT0 + 6 * j + k] = align ? r[k] : d[k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            y2[FEET0 + k] = c[FEET0 + k] * si[FEET0 + k] + mi[FEET0 + k];
            // after align_motions (201-d) smpl_to_ih appends the zero hand padding as "feet"
            y1[FEET0 + k] = align ? 0.f : (a[FEET0 + k] * sh[FEET0 + k] + mh[FEET0 + k]);
        }
    }

    V3 pos{a[3 * j] * sh[3 * j] + mh[3 * j], a[3 * j + 1] * sh[3 * j + 1] + mh[3 * j + 1], a[3 * j + 2] * sh[3 * j + 2] + mh[3 * j + 2]};
    V3 vel{a[66 + 3 * j] * sh[66 + 3 * j] + mh[66 + 3 * j], a[66 + 3 * j + 1] * sh[66 + 3 * j + 1] + mh[66 + 3 * j + 1],
           a[66 + 3 * j + 2] * sh[66 + 3 * j + 2] + mh[66 + 3 * j + 2]};
    if (align) {
        // align_motions(motion1 = interaction person (target), motion2 = individual person (moved))  alignment.py:112-158
        const float* a0 = o1 + seq;                               // frame 0
        const float* aL = o1 + seq + (size_t)(T - 1) * NF2;       // last frame (mask=None: alignment.py:86-88)
        const float* c0 = o2 + seq;
        const float* cL = o2 + seq + (size_t)(T - 1) * NF2;
        const V3 p1_0{c0[0] * si[0] + mi[0], c0[1] * si[1] + mi[1], c0[2] * si[2] + mi[2]};
        const V3 p1_L{cL[0] * si[0] + mi[0], cL[1] * si[1] + mi[1], cL[2] * si[2] + mi[2]};
        const V3 p2_0{a0[0] * sh[0] + mh[0], a0[1] * sh[1] + mh[1], a0[2] * sh[2] + mh[2]};
        const V3 p2_L{aL[0] * sh[0] + mh[0], aL[1] * sh[1] + mh[1], aL[2] * sh[2] + mh[2]};
        const V3 dl{p1_0.x - p2_0.x, p1_0.y - p2_0.y, p1_0.z - p2_0.z};
        const V3 t2_0{p2_0.x + dl.x, p2_0.y + dl.y, p2_0.z + dl.z};     // translated root, frame 0
        const V3 t2_L{p2_L.x + dl.x, p2_L.y + dl.y, p2_L.z + dl.z};
        V3 v1{p1_L.x - p1_0.x, 0.f, p1_L.z - p1_0.z};
        V3 v2{t2_L.x - t2_0.x, 0.f, t2_L.z - t2_0.z};
        const float n1 = sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + 1e-8f);
        const float n2 = sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + 1e-8f);
        v1 = V3{v1.x / n1, v1.y / n1, v1.z / n1};
        v2 = V3{v2.x / n2, v2.y / n2, v2.z / n2};
        const Q4 q = qbetween(v2, v1);
        const V3 r0 = qrot(q, t2_0);                                     // rotated root, frame 0
        const V3 d2{p1_0.x - r0.x, p1_0.y - r0.y, p1_0.z - r0.z};
        const V3 pr = qrot(q, V3{pos.x + dl.x, pos.y + dl.y, pos.z + dl.z});
        pos = V3{pr.x + d2.x, pr.y + d2.y, pr.z + d2.z};
        vel = qrot(q, vel);
    }
    y1[3 * j] = pos.x; y1[3 * j + 1] = pos.y; y1[3 * j + 2] = pos.z;
    y1[66 + 3 * j] = vel.x; y1[66 + 3 * j + 1] = vel.y; y1[66 + 3 * j + 2] = vel.z;
}

// ---------------------------------------------------------------------------------------------------------
// Influence expansion + blend + CFG combine.  One thread per (b, t, channel c in [0,524)).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int influence_index(int c) {   // 262 channels -> 23 groups  mixermdm.py:767-784
    return c < 66 ? c / 3 : (c < 132 ? (c - 66) / 3 : (c < FEET0 ? (c - ROT0) / 6 : 22));
}

// History side outputs under graph replay.  The destination pointers and the stride live in a DEVICE-side descriptor (mmdm_hist_desc,
// kernels.h) that the kernels read at run time, so a captured step graph does not depend on them: the same graph serves calls with
// different (or no) history buffers, and a stale buffer can never be baked into a replayed node.
// slot = *loop_pos / every when *loop_pos % every == 0, else none.
__device__ __forceinline__ long hist_slot(const int* loop_pos, int every) {
    if (!loop_pos) return 0;
    const int lp = *loop_pos;
    return (lp % every == 0) ? (long)(lp / every) : -1;
}

// hd != nullptr: history pointers / stride come from the descriptor (the explicit pointer arguments are ignored).
// nwh: channels of an influence history row: 262 (modes 3, 4: the expanded tensor) or 1 (modes 1, 2: the reference appends the
// un-expanded [2B, T, 1] tensor, mixermdm.py:739-745, 794-796).
template <bool RAG>
__global__ __launch_bounds__(256) void blend_cfg_kernel(const float* __restrict__ out1, const float* __restrict__ out2, const float* __restrict__ w,
                                                         int Tw, int nw, int use_force, float force, float s,
                                                         float* __restrict__ model_out, float* __restrict__ hist_i1, float* __restrict__ hist_i2,
                                                         float* __restrict__ hist_mix, const mmdm_hist_desc* __restrict__ hd,
                                                         const int* __restrict__ loop_pos, int nwh, int B, int T, mmdm_rag rg) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // frames of one half of the CFG-doubled batch: B * T, or the ragged batch's group stride (padding rows included: history slots are [2 * rows, C])
    const size_t half = RAG ? (size_t)rg.rows : (size_t)B * T;
    const size_t total = half * NF2;
    if (idx >= total) return;
    int every = 1;
    if (hd) { hist_i1 = hd->i1; hist_i2 = hd->i2; hist_mix = hd->mix; every = hd->every; }
    const long slot = hist_slot(loop_pos, every);
    if (slot < 0) { hist_i1 = nullptr; hist_i2 = nullptr; hist_mix = nullptr; }
    else {
        if (hist_i1) hist_i1 += (size_t)slot * 2 * half * nwh;
        if (hist_i2) hist_i2 += (size_t)slot * 2 * half * nwh;
        if (hist_mix) hist_mix += (size_t)slot * 2 * half * NF2;
    }
    const int ch = (int)(idx % NF2);
    const size_t fr = idx / NF2;                        // frame row inside the half
    int b;
    if constexpr (RAG) { b = rg.row_item[fr]; if (b < 0) return; }
    else b = (int)(fr / T);
    const int p = ch >= NF ? 1 : 0;
    const int c = ch - p * NF;
    const int wi = nw == 1 ? 0 : influence_index(c);
    float mix[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                       // u = 0: cond row b, u = 1: uncond row B + b
        const size_t frow = (size_t)u * half + fr;      // frame row of the doubled batch (= (b + u B) T + t in the uniform layout)
        const size_t e = frow * NF2 + ch;
        // w: [2 persons][2B sequences][Tw][nw] -- per frame (Tw = T: the Influence head's rows, person-major like the stack's) or per sequence (Tw = 1)
        float wv = Tw == 1 ? w[((size_t)p * 2 * B + (b + u * B)) * nw + wi] : w[((size_t)p * 2 * half + frow) * nw + wi];
        if (use_force) wv = 1.0f * force;
        const float v1 = out1[e], v2 = out2[e];
        mix[u] = v2 + wv * (v1 - v2);
        if (hist_mix) hist_mix[e] = mix[u];
        float* hi = p ? hist_i2 : hist_i1;
        if (hi && c < nwh) hi[frow * nwh + c] = wv;
    }
    model_out[idx] = s * mix[0] + (1.0f - s) * mix[1];
}

// ---------------------------------------------------------------------------------------------------------
// process_xstart + two-chain DDIM update.
// ---------------------------------------------------------------------------------------------------------
// floor[b, p] = min over (t, joint) of position Y   (center_motion "Put on Floor", alignment.py:182-184)
template <bool RAG>
__global__ __launch_bounds__(256) void floor_kernel(const float* __restrict__ m, float* __restrict__ floor_ws, int T, mmdm_rag rg) {
    const int bp = blockIdx.x;                   // b * 2 + p
    size_t row0 = (size_t)(bp >> 1) * T;
    if constexpr (RAG) { row0 = (size_t)rg.item_off[bp >> 1]; T = rg.item_len[bp >> 1]; }
    const float* base = m + row0 * NF2 + (size_t)(bp & 1) * NF;
    float v = INFINITY;
    for (int i = threadIdx.x; i < T * NJ; i += blockDim.x) {
        const int t = i / NJ, j = i % NJ;
        v = fminf(v, base[(size_t)t * NF2 + 3 * j + 1]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) floor_ws[bp] = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
}

__device__ __forceinline__ float ddim(float x, float x0, float c0, float c1, float c2, float c3) {
    const float eps = (c0 * x - x0) / c1;        // _predict_eps_from_xstart  gaussian_diffusion.py:558-562
    return x0 * c2 + c3 * eps;                   // eta = 0 mean             :1949-1956
}

template <bool RAG>
__global__ __launch_bounds__(256) void xstart_ddim_kernel(const float* __restrict__ m, const float* __restrict__ stats, const float* __restrict__ coef,
                                                           int S, const int* __restrict__ step_idx, float* __restrict__ x, float* __restrict__ x2,
                                                           float* __restrict__ px1, float* __restrict__ px2, const float* __restrict__ floor_ws,
                                                           int B, int T, int align, mmdm_rag rg) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int j, t, p, b;
    size_t seq;
    if constexpr (RAG) {                         // thread = (frame row, person, joint) of the ragged batch (mixer_pre_kernel)
        if (idx >= rg.rows * 2 * NJ) return;
        j = idx % NJ;
        p = (idx / NJ) % 2;
        const int r = idx / (NJ * 2);
        b = rg.row_item[r];
        if (b < 0) return;
        t = rg.row_pos[r];
        seq = (size_t)(r - t) * NF2 + (size_t)p * NF;
    } else {
        const int total = B * 2 * T * NJ;
        if (idx >= total) return;
        j = idx % NJ;
        t = (idx / NJ) % T;
        p = (idx / (NJ * T)) % 2;
        b = idx / (NJ * T * 2);
        seq = (size_t)b * T * NF2 + (size_t)p * NF;
    }
    const int i = *step_idx;
    const float c0 = coef[i], c1 = coef[S + i], c2 = coef[2 * S + i], c3 = coef[3 * S + i];
    const bool norm = i > 0;                                     // `if t[0] > 0`  gaussian_diffusion.py:2052
    const float* mh = stats, *sh = stats + NF, *mi = stats + 2 * NF, *si = stats + 3 * NF;
    const size_t off = seq + (size_t)t * NF2;
    const float* mo = m + off;

    // channels owned by this thread: pos j (3), vel j (3), rot j (6, j < 21) or feet (4, j == 21)
    float raw[12], v1[12];
    int ch[12];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { ch[cnt] = 3 * j + k; ++cnt; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { ch[cnt] = 66 + 3 * j + k; ++cnt; }
    if (j < 21) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { ch[cnt] = ROT0 + 6 * j + k; ++cnt; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ch[cnt] = FEET0 + k; ++cnt; }
        ch[10] = 0; ch[11] = 0;
    }
    const int nch = (j < 21) ? 12 : 10;
#pragma unroll
    for (int k = 0; k < 12; ++k) { raw[k] = (k < nch) ? mo[ch[k]] : 0.f; v1[k] = raw[k]; }

    if (norm && align) {
        // center_motion  alignment.py:161-222 on the ih_to_smpl'ed motion (positions / velocities unchanged by ih_to_smpl)
        const float fl = floor_ws[b * 2 + p];
        const float* f0 = m + seq;                                // frame 0
        const V3 root{f0[0], f0[1] - fl, f0[2]};
        const V3 rh{f0[3 * R_HIP], f0[3 * R_HIP + 1] - fl, f0[3 * R_HIP + 2]};
        const V3 lhp{f0[3 * L_HIP], f0[3 * L_HIP + 1] - fl, f0[3 * L_HIP + 2]};
        V3 ac{rh.x - lhp.x, rh.y - lhp.y, rh.z - lhp.z};
        const float an = sqrtf(ac.x * ac.x + ac.y * ac.y + ac.z * ac.z);
        ac = V3{ac.x / an, ac.y / an, ac.z / an};
        V3 fw = cross3(V3{0.f, 1.f, 0.f}, ac);
        const float fn = sqrtf(fw.x * fw.x + fw.y * fw.y + fw.z * fw.z);
        fw = V3{fw.x / fn, fw.y / fn, fw.z / fn};
        const Q4 q = qbetween(fw, V3{0.f, 0.f, 1.f});
        const V3 xz{root.x * 1.f, root.y * 0.f, root.z * 1.f};
        const V3 pc = qrot(q, V3{raw[0] - xz.x, (raw[1] - fl) - xz.y, raw[2] - xz.z});
        const V3 vc = qrot(q, V3{raw[3], raw[4], raw[5]});
        v1[0] = pc.x; v1[1] = pc.y; v1[2] = pc.z;
        v1[3] = vc.x; v1[4] = vc.y; v1[5] = vc.z;
        if (j < 21) {
            float d[6], r[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = raw[6 + k];
            rot6d_roundtrip(d, r);
#pragma unroll
            for (int k = 0; k < 6; ++k) v1[6 + k] = r[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v1[6 + k] = 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        if (k >= nch) continue;
        const int c = ch[k];
        float x0a = v1[k], x0b = raw[k];
        if (norm) {
            x0a = (x0a - mh[c]) / sh[c];
            x0b = (x0b - mi[c]) / si[c];
        }
        const size_t e = off + c;
        x[e] = ddim(x[e], x0a, c0, c1, c2, c3);
        x2[e] = ddim(x2[e], x0b, c0, c1, c2, c3);
        if (px1) px1[e] = x0a;
        if (px2) px2[e] = x0b;
    }
}

__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* __restrict__ m, const float* __restrict__ coef, int S, const int* __restrict__ step_idx,
                                                        float s, float* __restrict__ x, float* __restrict__ px, size_t per_batch_total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_batch_total) return;
    const int i = *step_idx;
    const float x0 = s * m[idx] + (1.0f - s) * m[per_batch_total + idx];   // cfg_sampler.py:24-28
    x[idx] = ddim(x[idx], x0, coef[i], coef[S + i], coef[2 * S + i], coef[3 * S + i]);
    if (px) px[idx] = x0;
}

__global__ __launch_bounds__(256) void cfg4_ddim_kernel(const float* __restrict__ m, const float* __restrict__ coef, int S, const int* __restrict__ step_idx,
                                                         float s, float si, float sd, float* __restrict__ x, float* __restrict__ px, size_t per_batch_total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_batch_total) return;
    const int i = *step_idx;
    // (s*full) + (s_int*interaction-only) + (s_ind*individuals-only) + ((1-(s+s_int+s_ind))*uncond)   cfg_sampler.py:97
    const float x0 = (s * m[idx]) + (si * m[per_batch_total + idx]) + (sd * m[2 * per_batch_total + idx]) + ((1.0f - (s + si + sd)) * m[3 * per_batch_total + idx]);
    x[idx] = ddim(x[idx], x0, coef[i], coef[S + i], coef[2 * S + i], coef[3 * S + i]);
    if (px) px[idx] = x0;
}

// ClassifierFreeSampleDualMDM.forward combine (cfg_sampler.py:139-150) + DDIM update: per-model guidance, then the
// time-scheduled blend out_int + w[step] (out_ind - out_int).
__global__ __launch_bounds__(256) void dual_ddim_kernel(const float* __restrict__ mi, const float* __restrict__ mI, const float* __restrict__ coef, int S,
                                                         const int* __restrict__ step_idx, const float* __restrict__ wtab, float s_ind, float s_int,
                                                         float* __restrict__ x, float* __restrict__ px, size_t per_batch_total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_batch_total) return;
    const int i = *step_idx;
    const float ui = mi[per_batch_total + idx], uI = mI[per_batch_total + idx];
    const float gI = uI + s_int * (mI[idx] - uI);
    const float gi = ui + s_ind * (mi[idx] - ui);
    const float x0 = gI + wtab[i] * (gi - gI);
    x[idx] = ddim(x[idx], x0, coef[i], coef[S + i], coef[2 * S + i], coef[3 * S + i]);
    if (px) px[idx] = x0;
}

__global__ void step_dec_kernel(int* step_idx, int* loop_pos) { *step_idx -= 1; *loop_pos += 1; }
__global__ void set_step_kernel(int* step_idx, int* loop_pos, int s, int l) { *step_idx = s; *loop_pos = l; }

// which: 0 -> hd->o1, 1 -> hd->o2; a null destination (history not requested for this call) makes the launch a no-op
__global__ void hist_copy_kernel(const float* __restrict__ src, const mmdm_hist_desc* __restrict__ hd, int which, size_t count, const int* __restrict__ loop_pos) {
    float* dst = which ? hd->o2 : hd->o1;
    if (!dst) return;
    const long slot = hist_slot(loop_pos, hd->every);
    if (slot < 0) return;
    float* d = dst + (size_t)slot * count;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) d[i] = src[i];
}

// Pose rows -> 16-byte-aligned, zero-padded GEMM operands: dst[p][r][0:ldp] = (src[r][p*262 : p*262+262], 0 ...).  The 262-float person
// slices of a [rows, 524] motion tensor start at byte 1048 (not 16-byte aligned) and 262 is not a multiple of the GEMM's K step, so
// the embedding GEMMs would otherwise run on the register-staged fallback kernel.
__global__ void repack_pose_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int npers, int rows, int ldp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)rows * ldp;
    if (i >= per * npers) return;
    const int p = (int)(i / per);
    const size_t rem = i - (size_t)p * per;
    const size_t r = rem / ldp;
    const int c = (int)(rem - r * ldp);
    dst[i] = c < MMDM_NF ? src[r * ld_src + (size_t)p * MMDM_NF + c] : 0.f;
}

// the same rows as the two fp16 planes of the fp32-split GEMM (kernels.h mmdm_split2): dst [npers][2][rows][ldp]
__global__ void repack_pose_split_kernel(const float* __restrict__ src, int ld_src, _Float16* __restrict__ dst, int npers, int rows, int ldp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)rows * ldp;
    if (i >= per * npers) return;
    const int p = (int)(i / per);
    const size_t rem = i - (size_t)p * per;
    const size_t r = rem / ldp;
    const int c = (int)(rem - r * ldp);
    const float x = c < MMDM_NF ? src[r * ld_src + (size_t)p * MMDM_NF + c] : 0.f;
    _Float16 h, l;
    mmdm_split2(x, h, l);
    dst[(size_t)p * 2 * per + rem] = h;
    dst[(size_t)p * 2 * per + per + rem] = l;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int n, int D) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * D) return;
    dst[i] = src[(size_t)idx[i / D] * D + (i % D)];
}

}  // namespace

extern "C" int mmdm_mixer_pre_f32(const float* o1, const float* o2, const float* stats, float* out1, float* out2,
                                  int n, int T, int align, void* stream) {
    if (n == 0 || T == 0) return MMDM_OK;
    if (!o1 || !o2 || !stats || !out1 || !out2 || n < 0 || T < 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_mixer_pre_f32: bad arguments");
    const int total = n * 2 * T * MMDM_NJ;
    hipLaunchKernelGGL(mixer_pre_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), o1, o2, stats, out1, out2, n, T, align, mmdm_rag{});
    return mmdm_check_launch("mixer_pre");
}

// ragged batch: `groups` groups of rg.rows frame rows (the cond | uncond halves of the CFG-doubled batch)
int mmdm_mixer_pre_rag(const float* o1, const float* o2, const float* stats, float* out1, float* out2, int groups, int align, const mmdm_rag& rg, hipStream_t st) {
    const int total = groups * rg.rows * 2 * MMDM_NJ;
    hipLaunchKernelGGL(mixer_pre_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, st, o1, o2, stats, out1, out2, groups, 0, align, rg);
    return mmdm_check_launch("mixer_pre_rag");
}

extern "C" int mmdm_blend_cfg_f32(const float* out1, const float* out2, const float* w, int mode, int use_force, float force,
                                  float cfg_scale, float* model_out, float* hist_i1, float* hist_i2, float* hist_mix,
                                  int B, int T, void* stream) {
    if (B == 0 || T == 0) return MMDM_OK;
    if (!out1 || !out2 || !w || !model_out || B < 0 || T < 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_blend_cfg_f32: bad arguments");
    if (mode < 1 || mode > 4) return mmdm_set_error(MMDM_ERR_ARG, "Mixing mode not recognized");   // mixermdm.py:786
    const int Tw = (mode == 2 || mode == 4) ? T : 1;
    const int nw = (mode >= 3) ? 23 : 1;
    const size_t total = (size_t)B * T * NF2;
    hipLaunchKernelGGL(blend_cfg_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       out1, out2, w, Tw, nw, use_force, force, cfg_scale, model_out, hist_i1, hist_i2, hist_mix,
                       (const mmdm_hist_desc*)nullptr, (const int*)nullptr, MMDM_NF, B, T, mmdm_rag{});
    return mmdm_check_launch("blend_cfg");
}

// Same as mmdm_blend_cfg_f32 with the history destinations read from the device-side descriptor and the slot chosen on the device
// (graph replay): see hist_slot().  Influence history rows are [.., 262] for modes 3-4 and [.., 1] for modes 1-2 (the reference's shapes).
int mmdm_blend_cfg_dyn(const float* out1, const float* out2, const float* w, int mode, int use_force, float force, float cfg_scale,
                       float* model_out, const mmdm_hist_desc* hd, const int* loop_pos, int B, int T, hipStream_t st) {
    if (mode < 1 || mode > 4) return mmdm_set_error(MMDM_ERR_ARG, "Mixing mode not recognized");
    const int Tw = (mode == 2 || mode == 4) ? T : 1;
    const int nw = (mode >= 3) ? 23 : 1;
    const size_t total = (size_t)B * T * NF2;
    hipLaunchKernelGGL(blend_cfg_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       out1, out2, w, Tw, nw, use_force, force, cfg_scale, model_out, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       hd, loop_pos, mode >= 3 ? MMDM_NF : 1, B, T, mmdm_rag{});
    return mmdm_check_launch("blend_cfg");
}

// ragged batch: history slots are [2 * rg.rows, C] (padding rows are never written)
int mmdm_blend_cfg_rag(const float* out1, const float* out2, const float* w, int mode, int use_force, float force, float cfg_scale,
                       float* model_out, const mmdm_hist_desc* hd, const int* loop_pos, const mmdm_rag& rg, hipStream_t st) {
    if (mode < 1 || mode > 4) return mmdm_set_error(MMDM_ERR_ARG, "Mixing mode not recognized");
    const int Tw = (mode == 2 || mode == 4) ? 2 : 1;        // (only "per frame" vs "per sequence" matters to the ragged form)
    const int nw = (mode >= 3) ? 23 : 1;
    const size_t total = (size_t)rg.rows * NF2;
    hipLaunchKernelGGL(blend_cfg_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       out1, out2, w, Tw, nw, use_force, force, cfg_scale, model_out, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       hd, loop_pos, mode >= 3 ? MMDM_NF : 1, rg.B, 0, rg);
    return mmdm_check_launch("blend_cfg_rag");
}

// ---------------------------------------------------------------------------------------------------------
// Ragged batches: the row maps of one sampling call, built ON THE DEVICE from the item lengths (passed by value: no host buffer to keep
// alive, no synchronisation) -- item offsets / lengths, item and frame index of every frame row of a group (-1 / 0 for the padding rows
// between the sum of lengths and the group stride `rows`), and per sequence of the `groups` x B sequences its first row and length.
// ---------------------------------------------------------------------------------------------------------
struct RagLens { int v[MMDM_RAG_MAX_ITEMS]; };
__global__ __launch_bounds__(256) void rag_setup_kernel(RagLens L, int B, int rows, int groups, int* __restrict__ item_off, int* __restrict__ item_len,
                                                        int* __restrict__ row_item, int* __restrict__ row_pos, int* __restrict__ row_seq,
                                                        int* __restrict__ seq_off, int* __restrict__ seq_len, int* __restrict__ item_order) {
    __shared__ int off[MMDM_RAG_MAX_ITEMS + 1];
    if (threadIdx.x == 0) {
        int a = 0;
        for (int b = 0; b < B; ++b) { off[b] = a; a += L.v[b]; }
        off[B] = a;
    }
    __syncthreads();
    const int RT = off[B];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int r = tid; r < groups * rows; r += nth) {
        const int g = r / rows, rl = r % rows;
        int item = -1, pos = 0;
        if (rl < RT) {
            int lo = 0, hi = B - 1;                       // largest b with off[b] <= rl
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= rl) lo = mid; else hi = mid - 1; }
            item = lo; pos = rl - off[lo];
        }
        if (g == 0) { row_item[rl] = item; row_pos[rl] = pos; }
        row_seq[r] = g * B + (item < 0 ? 0 : item);
    }
    for (int s = tid; s < groups * B; s += nth) { seq_off[s] = (s / B) * rows + off[s % B]; seq_len[s] = L.v[s % B]; }
    for (int b = tid; b < B; b += nth) { item_off[b] = off[b]; item_len[b] = L.v[b]; }
    // items by length, longest first (ties in index order): item b's rank = the number of items that come before it (B <= 256: a rank sort by one block)
    if (blockIdx.x == 0 && item_order) {
        for (int b = threadIdx.x; b < B; b += blockDim.x) {
            int rank = 0;
            for (int c = 0; c < B; ++c) rank += (L.v[c] > L.v[b] || (L.v[c] == L.v[b] && c < b)) ? 1 : 0;
            item_order[rank] = b;
        }
    }
}

int mmdm_rag_setup(const int* lens_host, int B, int rows, int groups, int* item_off, int* item_len, int* row_item, int* row_pos, int* row_seq,
                   int* seq_off, int* seq_len, int* item_order, hipStream_t st) {
    if (B <= 0 || B > MMDM_RAG_MAX_ITEMS) return mmdm_set_error(MMDM_ERR_ARG, "ragged batch: %d items (1 .. %d)", B, MMDM_RAG_MAX_ITEMS);
    RagLens L;
    for (int b = 0; b < MMDM_RAG_MAX_ITEMS; ++b) L.v[b] = b < B ? lens_host[b] : 0;
    hipLaunchKernelGGL(rag_setup_kernel, dim3(64), dim3(256), 0, st, L, B, rows, groups, item_off, item_len, row_item, row_pos, row_seq, seq_off, seq_len, item_order);
    return mmdm_check_launch("rag_setup");
}

int mmdm_hist_copy(const float* src, const mmdm_hist_desc* hd, int which, size_t count, const int* loop_pos, hipStream_t st) {
    hipLaunchKernelGGL(hist_copy_kernel, dim3(1024), dim3(256), 0, st, src, hd, which, count, loop_pos);
    return mmdm_check_launch("hist_copy");
}

__global__ void set_hist_kernel(mmdm_hist_desc* d, mmdm_hist_desc v) { *d = v; }

int mmdm_set_hist_desc(mmdm_hist_desc* d, const mmdm_hist_desc& v, hipStream_t st) {
    hipLaunchKernelGGL(set_hist_kernel, dim3(1), dim3(1), 0, st, d, v);
    return mmdm_check_launch("set_hist");
}

int mmdm_set_step(int* step_idx, int* loop_pos, int s, int l, hipStream_t st) {
    hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(1), 0, st, step_idx, loop_pos, s, l);
    return mmdm_check_launch("set_step");
}

int mmdm_gather_rows(const float* src, const int* idx, float* dst, int n, int D, hipStream_t st) {
    const size_t total = (size_t)n * D;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, idx, dst, n, D);
    return mmdm_check_launch("gather_rows");
}

extern "C" int mmdm_xstart_ddim_f32(const float* model_out, const float* stats, const float* coef, int S, const int* step_idx,
                                    float* x, float* x2, float* pred_xstart, float* pred_xstart2, float* floor_ws,
                                    int B, int T, int align, void* stream) {
    if (B == 0 || T == 0) return MMDM_OK;
    if (!model_out || !stats || !coef || !step_idx || !x || !x2 || !floor_ws || S <= 0 || B < 0 || T < 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_xstart_ddim_f32: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (align) {
        hipLaunchKernelGGL(floor_kernel<false>, dim3(B * 2), dim3(256), 0, st, model_out, floor_ws, T, mmdm_rag{});
        if (int rc = mmdm_check_launch("floor")) return rc;
    }
    const int total = B * 2 * T * MMDM_NJ;
    hipLaunchKernelGGL(xstart_ddim_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, st, model_out, stats, coef, S, step_idx, x, x2,
                       pred_xstart, pred_xstart2, floor_ws, B, T, align, mmdm_rag{});
    return mmdm_check_launch("xstart_ddim");
}

// ragged batch (rg.B items, rg.rows frame rows incl. padding): same arithmetic per item
int mmdm_xstart_ddim_rag(const float* model_out, const float* stats, const float* coef, int S, const int* step_idx,
                         float* x, float* x2, float* pred_xstart, float* pred_xstart2, float* floor_ws, int align, const mmdm_rag& rg, hipStream_t st) {
    if (align) {
        hipLaunchKernelGGL(floor_kernel<true>, dim3(rg.B * 2), dim3(256), 0, st, model_out, floor_ws, 0, rg);
        if (int rc = mmdm_check_launch("floor_rag")) return rc;
    }
    const int total = rg.rows * 2 * MMDM_NJ;
    hipLaunchKernelGGL(xstart_ddim_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, st, model_out, stats, coef, S, step_idx, x, x2,
                       pred_xstart, pred_xstart2, floor_ws, rg.B, 0, align, rg);
    return mmdm_check_launch("xstart_ddim_rag");
}

extern "C" int mmdm_cfg_ddim_f32(const float* m, const float* coef, int S, const int* step_idx, float cfg_scale,
                                 float* x, float* pred_xstart, int B, int T, int C, void* stream) {
    if (B == 0 || T == 0) return MMDM_OK;
    if (!m || !coef || !step_idx || !x || S <= 0 || B < 0 || T < 0 || C <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_cfg_ddim_f32: bad arguments");
    const size_t total = (size_t)B * T * C;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       m, coef, S, step_idx, cfg_scale, x, pred_xstart, total);
    return mmdm_check_launch("cfg_ddim");
}

extern "C" int mmdm_cfg4_ddim_f32(const float* m, const float* coef, int S, const int* step_idx, float s, float s_int, float s_ind,
                                  float* x, float* pred_xstart, int B, int T, int C, void* stream) {
    if (B == 0 || T == 0) return MMDM_OK;
    if (!m || !coef || !step_idx || !x || S <= 0 || B < 0 || T < 0 || C <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_cfg4_ddim_f32: bad arguments");
    const size_t total = (size_t)B * T * C;
    hipLaunchKernelGGL(cfg4_ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       m, coef, S, step_idx, s, s_int, s_ind, x, pred_xstart, total);
    return mmdm_check_launch("cfg4_ddim");
}

extern "C" int mmdm_dual_ddim_f32(const float* m_ind, const float* m_int, const float* coef, int S, const int* step_idx, const float* w_table,
                                  float s_ind, float s_int, float* x, float* pred_xstart, int B, int T, int C, void* stream) {
    if (B == 0 || T == 0) return MMDM_OK;
    if (!m_ind || !m_int || !coef || !step_idx || !w_table || !x || S <= 0 || B < 0 || T < 0 || C <= 0)
        return mmdm_set_error(MMDM_ERR_ARG, "mmdm_dual_ddim_f32: bad arguments");
    const size_t total = (size_t)B * T * C;
    hipLaunchKernelGGL(dual_ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       m_ind, m_int, coef, S, step_idx, w_table, s_ind, s_int, x, pred_xstart, total);
    return mmdm_check_launch("dual_ddim");
}

// split != 0: dst receives [npers][2 planes][rows][ldp] fp16 (the A operand of mmdm_linear_split) instead of [npers][rows][ldp] fp32
int mmdm_repack_pose(const float* src, int ld_src, float* dst, int npers, int rows, int ldp, int split, hipStream_t st) {
    const size_t total = (size_t)npers * rows * ldp;
    if (total == 0) return MMDM_OK;
    if (split) hipLaunchKernelGGL(repack_pose_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld_src, reinterpret_cast<_Float16*>(dst), npers, rows, ldp);
    else hipLaunchKernelGGL(repack_pose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, ld_src, dst, npers, rows, ldp);
    return mmdm_check_launch("repack_pose");
}

extern "C" int mmdm_gather_rows_f32(const float* src, const int* idx, float* dst, int n, int D, void* stream) {
    if (n == 0) return MMDM_OK;
    if (!src || !idx || !dst || n < 0 || D <= 0) return mmdm_set_error(MMDM_ERR_ARG, "mmdm_gather_rows_f32: bad arguments");
    return mmdm_gather_rows(src, idx, dst, n, D, static_cast<hipStream_t>(stream));
}

int mmdm_step_dec(int* step_idx, int* loop_pos, hipStream_t st) {
    hipLaunchKernelGGL(step_dec_kernel, dim3(1), dim3(1), 0, st, step_idx, loop_pos);
    return mmdm_check_launch("step_dec");
}
