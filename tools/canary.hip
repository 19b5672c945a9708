// Canary (GPU box): a kernel that does nothing but hold known values -- in LDS, in registers, in its own global buffer, and in LDS through the
// LDS-DMA path the library's GEMMs use -- and re-checks them for a few milliseconds while something else runs beside it.  Built as a small
// shared object for tools/overlap_bisect.py (MODE=canary):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/canary.hip -o build/libcanary.so
// report[0..3] = mismatches seen in (LDS, registers, global memory, LDS-DMA image); report[4] = checks made.
// The other kernels below are the ARITHMETIC canaries that found round 5's gfx950 hazard (LAB_NOTES.md): canary_trans_kernel (the rotation round trip of
// csrc/geometry.hip on fixed inputs, bit-compared with the same thread's first result), canary_ops_kernel (single operations), canary_chain_kernel (the first
// intermediate that moves) and aggressor_kernel (micro-aggressors).  Variants: -Xclang -target-feature -Xclang -packed-fp32-ops builds the canaries that never
// move; the hand-assembled route (wait states behind every v_pk_*_f32: they change nothing) is
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S tools/canary.hip -o c.s;  <insert s_nop N behind each v_pk_*_f32>;
//   clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c c.s -o c.o;  ld.lld -shared c.o -o build/canary_nopN.hsaco;  CANARY_HSACO=build/canary_nopN.hsaco
#include <hip/hip_runtime.h>

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0xC2B2AE3Du; x ^= x >> 13;
    return x;
}

__global__ __launch_bounds__(256) void canary_kernel(unsigned* report, unsigned* gbuf, const unsigned* pattern, int lds_words, int spin_us) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const int tid = threadIdx.x, wg = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lds_words / 2;                       // first half: plain LDS pattern; second half: LDS-DMA image of `pattern`
    unsigned regs[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) regs[i] = mix(wg * 256 + tid, i);
    for (int i = tid; i < half; i += 256) smem[i] = mix(wg, i);
    unsigned* mine = gbuf + (size_t)wg * 4096;
    for (int i = tid; i < 4096; i += 256) mine[i] = mix(wg + 77, i);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(pattern), 0, 0xffffffff, 0x00020000);
    const int pieces = half / 256;                        // 1-KiB pieces (64 lanes x 16 bytes) of the DMA half
    unsigned bad_lds = 0, bad_reg = 0, bad_glb = 0, bad_dma = 0, checks = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __shared__ int stop;
    __syncthreads();
    for (;;) {
        if (tid == 0) stop = __builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)spin_us * 100;      // one decision for the whole workgroup
        __syncthreads();
        if (stop) break;
        // LDS-DMA: this wave's pieces of the pattern into the second half, as the GEMMs stage their operands
        for (int pc = wave; pc < pieces; pc += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + half + pc * 256), 16, lane * 16, ((pc + checks) % 64) * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < half; i += 256) bad_lds += smem[i] != mix(wg, i);
        for (int pc = 0; pc < pieces; ++pc) {
            const unsigned want = pattern[((pc + checks) % 64) * 256 + tid];
            bad_dma += smem[half + pc * 256 + tid] != want;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("" : "+v"(regs[i]));
            bad_reg += regs[i] != mix(wg * 256 + tid, i);
        }
        for (int i = tid; i < 4096; i += 256) bad_glb += __builtin_nontemporal_load(mine + i) != mix(wg + 77, i);
        ++checks;
        __syncthreads();
    }
    if (bad_lds) atomicAdd(report + 0, bad_lds);
    if (bad_reg) atomicAdd(report + 1, bad_reg);
    if (bad_glb) atomicAdd(report + 2, bad_glb);
    if (bad_dma) atomicAdd(report + 3, bad_dma);
    if (tid == 0) atomicAdd(report + 4, checks);
#endif
}

// The arithmetic canary: the rotation round trip of csrc/geometry.hip (sqrt, divisions, atan2f, sinf, cosf) on fixed inputs, recomputed over and
// over and compared bit for bit with the same thread's first result.  report[5] = lanes whose result moved, report[6] = evaluations.
__device__ __forceinline__ float c_sqrt_pos(float x) { return x > 0.f ? sqrtf(x) : 0.f; }
__device__ __forceinline__ float c_copysign(float a, float b) { return ((a < 0.f) != (b < 0.f)) ? -a : a; }
__device__ __forceinline__ float c_half_sinc(float half, float ang) { return (fabsf(ang) < 1e-6f) ? (0.5f - (ang * ang) / 48.f) : (sinf(half) / ang); }
__device__ __forceinline__ void c_roundtrip(const float d[6], float out[6]) {
    float a1x = d[0], a1y = d[2], a1z = d[4], a2x = d[1], a2y = d[3], a2z = d[5];
    float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
    float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
    const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    const float m00 = b1x, m01 = b1y, m02 = b1z, m10 = b2x, m11 = b2y, m12 = b2z, m20 = b3x, m21 = b3y, m22 = b3z;
    const float qw = 0.5f * c_sqrt_pos(1.f + m00 + m11 + m22);
    const float qx = c_copysign(0.5f * c_sqrt_pos(1.f + m00 - m11 - m22), m21 - m12);
    const float qy = c_copysign(0.5f * c_sqrt_pos(1.f - m00 + m11 - m22), m02 - m20);
    const float qz = c_copysign(0.5f * c_sqrt_pos(1.f - m00 - m11 + m22), m10 - m01);
    const float nrm = sqrtf(qx * qx + qy * qy + qz * qz);
    const float half = atan2f(nrm, qw);
    const float ang = 2.f * half;
    const float s1 = c_half_sinc(half, ang);
    const float ax = qx / s1, ay = qy / s1, az = qz / s1;
    const float ang2 = sqrtf(ax * ax + ay * ay + az * az);
    const float half2 = 0.5f * ang2;
    const float s2 = c_half_sinc(half2, ang2);
    const float r = cosf(half2), i = ax * s2, j = ay * s2, k = az * s2;
    const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
    out[0] = 1.f - two_s * (j * j + k * k); out[2] = two_s * (i * j - k * r); out[4] = two_s * (i * k + j * r);
    out[1] = two_s * (i * j + k * r); out[3] = 1.f - two_s * (i * i + k * k); out[5] = two_s * (j * k - i * r);
}

__global__ __launch_bounds__(256) void canary_trans_kernel(unsigned* report, const float* inputs, int spin_us) {
    // report[5] evaluations that moved, [6] evaluations, [7] of the moved ones: the INPUT registers no longer hold the inputs,
    // [20] the REFERENCE registers no longer hold the first result, [21] neither (the arithmetic itself gave other bits)
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    __shared__ float keep_d[6][256], keep_r[6][256];
    float d[6], ref[6], out[6];
    for (int k = 0; k < 6; ++k) { d[k] = inputs[(size_t)gid * 6 + k]; keep_d[k][tid] = d[k]; }
    for (int k = 0; k < 6; ++k) ref[k] = 0.f;
    __shared__ int stop;
    unsigned moved = 0, evals = 0, bad_in = 0, bad_ref = 0, bad_calc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (tid == 0) stop = __builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)spin_us * 100;
        __syncthreads();
        if (stop) break;
        for (int rep = 0; rep < 8; ++rep) {
            for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(d[k]));
            c_roundtrip(d, out);
            if (evals == 0) {                    // the reference comes from the SAME instruction sequence (an inlined second copy may contract differently)
                for (int k = 0; k < 6; ++k) { ref[k] = out[k]; keep_r[k][tid] = out[k]; }
            } else {
                bool m = false;
                for (int k = 0; k < 6; ++k) m = m || __float_as_uint(out[k]) != __float_as_uint(ref[k]);
                if (m) {
                    bool din = false, dref = false;
                    for (int k = 0; k < 6; ++k) { din = din || __float_as_uint(d[k]) != __float_as_uint(keep_d[k][tid]); dref = dref || __float_as_uint(ref[k]) != __float_as_uint(keep_r[k][tid]); }
                    ++moved; bad_in += din; bad_ref += dref; bad_calc += !din && !dref;
                }
            }
            ++evals;
        }
        __syncthreads();
    }
    if (moved) { atomicAdd(report + 5, moved); atomicAdd(report + 7, bad_in); atomicAdd(report + 20, bad_ref); atomicAdd(report + 21, bad_calc); }
    if (tid == 0) atomicAdd(report + 6, evals);
}

// Where in the round trip do the bits first move?  Every intermediate of the chain is compared with the first evaluation's; report[32 + i] counts the
// evaluations whose FIRST moved intermediate was number i (order: b1 xyz 0-2, dt 3, b2 xyz 4-6, b3 xyz 7-9, qw 10, qx 11, qy 12, qz 13, nrm 14, half 15,
// s1 16, ax ay az 17-19, ang2 20, s2 21, r 22, i j k 23-25, two_s 26, out 27-32)
__global__ __launch_bounds__(256) void canary_chain_kernel(unsigned* report, const float* inputs, int spin_us) {
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    constexpr int NI = 33;
    float d[6];
    for (int k = 0; k < 6; ++k) d[k] = inputs[(size_t)gid * 6 + k];
    unsigned ref[NI];
    for (int k = 0; k < NI; ++k) ref[k] = 0;
    __shared__ int stop;
    __shared__ unsigned first_moved[NI];
    if (tid < NI) first_moved[tid] = 0;
    unsigned evals = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (tid == 0) stop = __builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)spin_us * 100;
        __syncthreads();
        if (stop) break;
        for (int rep = 0; rep < 4; ++rep) {
            for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(d[k]));
            float v[NI];
            float a1x = d[0], a1y = d[2], a1z = d[4], a2x = d[1], a2y = d[3], a2z = d[5];
            float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
            const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
            v[0] = b1x; v[1] = b1y; v[2] = b1z;
            const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
            v[3] = dt;
            float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
            const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
            b2x /= n2; b2y /= n2; b2z /= n2;
            v[4] = b2x; v[5] = b2y; v[6] = b2z;
            const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
            v[7] = b3x; v[8] = b3y; v[9] = b3z;
            const float m00 = b1x, m01 = b1y, m02 = b1z, m10 = b2x, m11 = b2y, m12 = b2z, m20 = b3x, m21 = b3y, m22 = b3z;
            const float qw = 0.5f * c_sqrt_pos(1.f + m00 + m11 + m22);
            const float qx = c_copysign(0.5f * c_sqrt_pos(1.f + m00 - m11 - m22), m21 - m12);
            const float qy = c_copysign(0.5f * c_sqrt_pos(1.f - m00 + m11 - m22), m02 - m20);
            const float qz = c_copysign(0.5f * c_sqrt_pos(1.f - m00 - m11 + m22), m10 - m01);
            v[10] = qw; v[11] = qx; v[12] = qy; v[13] = qz;
            const float nrm = sqrtf(qx * qx + qy * qy + qz * qz);
            const float half = atan2f(nrm, qw);
            v[14] = nrm; v[15] = half;
            const float ang = 2.f * half;
            const float s1 = c_half_sinc(half, ang);
            v[16] = s1;
            const float ax = qx / s1, ay = qy / s1, az = qz / s1;
            v[17] = ax; v[18] = ay; v[19] = az;
            const float ang2 = sqrtf(ax * ax + ay * ay + az * az);
            const float half2 = 0.5f * ang2;
            const float s2 = c_half_sinc(half2, ang2);
            v[20] = ang2; v[21] = s2;
            const float r = cosf(half2), i = ax * s2, j = ay * s2, k = az * s2;
            v[22] = r; v[23] = i; v[24] = j; v[25] = k;
            const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
            v[26] = two_s;
            v[27] = 1.f - two_s * (j * j + k * k); v[29] = two_s * (i * j - k * r); v[31] = two_s * (i * k + j * r);
            v[28] = two_s * (i * j + k * r); v[30] = 1.f - two_s * (i * i + k * k); v[32] = two_s * (j * k - i * r);
            if (evals == 0) { for (int q = 0; q < NI; ++q) ref[q] = __float_as_uint(v[q]); }
            else {
                int fm = -1;
                for (int q = NI - 1; q >= 0; --q) if (__float_as_uint(v[q]) != ref[q]) fm = q;
                if (fm >= 0) atomicAdd(&first_moved[fm], 1u);
            }
            ++evals;
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid < NI && first_moved[tid]) atomicAdd(report + 32 + tid, first_moved[tid]);
    if (tid == 0) atomicAdd(report + 6, evals);
}

extern "C" __attribute__((visibility("default"))) int canary_chain_launch(unsigned* report, const float* inputs, int nwg, int spin_us, void* stream) {
    hipLaunchKernelGGL(canary_chain_kernel, dim3(nwg), dim3(256), 0, static_cast<hipStream_t>(stream), report, inputs, spin_us);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// WHAT is the wrong result?  One explicit `v_pk_mul_f32 d, a, b op_sel_hi:[1,0]` (d.lo = a.lo * b.lo, d.hi = a.hi * b.lo) per iteration, its destination pre-set to a
// sentinel pair, between a few dependent plain FMAs (dense VALU code around it, as in the round trip).  A mismatch is recorded with its operands:
// rec[8 i ..] = a.lo, a.hi, b.lo, b.hi, d.lo, d.hi, which lane (1 lo, 2 hi, 3 both), iteration.  report[5] mismatches, report[6] evaluations (per workgroup).
typedef float f32x2c __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void canary_pk_kernel(unsigned* report, float* rec, const float* inputs, int spin_us) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    f32x2c a{inputs[(size_t)gid * 6], inputs[(size_t)gid * 6 + 1]}, b{inputs[(size_t)gid * 6 + 2], inputs[(size_t)gid * 6 + 3]};
    float f = inputs[(size_t)gid * 6 + 4], g = inputs[(size_t)gid * 6 + 5];
    __shared__ int stop;
    unsigned evals = 0, bad = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (tid == 0) stop = __builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)spin_us * 100;
        __syncthreads();
        if (stop) break;
        for (int rep = 0; rep < 16; ++rep) {
            asm volatile("" : "+v"(a), "+v"(b), "+v"(f), "+v"(g));
            float want_lo, want_hi;                                   // by the plain multiply, in asm (the compiler would form the same packed instruction again)
            { const float a0 = a[0], a1 = a[1], b0 = b[0];
              asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a0), "v"(a1), "v"(b0)); }
            f32x2c d{777.0f, 888.0f};
            float t1 = f, t2 = g;
            asm volatile("v_fma_f32 %1, %1, %5, %6\n\t"
                         "v_fma_f32 %2, %2, %6, %5\n\t"
                         "v_pk_mul_f32 %0, %3, %4 op_sel_hi:[1,0]\n\t"
                         "v_fma_f32 %1, %1, %2, %5\n\t"
                         "v_fma_f32 %2, %2, %1, %6"
                         : "+v"(d), "+v"(t1), "+v"(t2) : "v"(a), "v"(b), "v"(f), "v"(g));
            const bool blo = __float_as_uint(d[0]) != __float_as_uint(want_lo), bhi = __float_as_uint(d[1]) != __float_as_uint(want_hi);
            if (blo || bhi) {
                const unsigned slot = atomicAdd(report + 5, 1u);
                if (slot < 256) {
                    float* r = rec + 8 * slot;
                    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = d[0]; r[5] = d[1]; r[6] = (float)((blo ? 1 : 0) | (bhi ? 2 : 0)); r[7] = (float)evals;
                }
                ++bad;
            }
            if (t1 == 12345.5f && t2 == 54321.5f) f += 1.0f;          // keep the FMAs alive
            ++evals;
        }
        __syncthreads();
    }
    if (tid == 0) atomicAdd(report + 6, evals);
#endif
}

extern "C" __attribute__((visibility("default"))) int canary_pk_launch(unsigned* report, float* rec, const float* inputs, int nwg, int spin_us, void* stream) {
    hipLaunchKernelGGL(canary_pk_kernel, dim3(nwg), dim3(256), 0, static_cast<hipStream_t>(stream), report, rec, inputs, spin_us);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Which operation moves?  One op per slot on fixed per-thread inputs, compared bit for bit with the same sequence's first result.
// report[8 + slot] = evaluations that moved: 0 fma chain, 1 division, 2 sqrtf, 3 sinf, 4 cosf, 5 atan2f, 6 expf, 7 erff, 8 integer mix, 9 v_rcp_f32, 10 v_sin_f32 (native), 11-13 packed-fp32 fma / mul / add
__global__ __launch_bounds__(256) void canary_ops_kernel(unsigned* report, const float* inputs, int spin_us) {
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    constexpr int NOP = 14;
    float a = inputs[(size_t)gid * 6], b = inputs[(size_t)gid * 6 + 1] + 3.0f, c = inputs[(size_t)gid * 6 + 2];
    unsigned ref[NOP], moved[NOP];
    for (int k = 0; k < NOP; ++k) { ref[k] = 0; moved[k] = 0; }
    __shared__ int stop;
    unsigned evals = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (tid == 0) stop = __builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)spin_us * 100;
        __syncthreads();
        if (stop) break;
        for (int rep = 0; rep < 4; ++rep) {
            asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
            unsigned r[NOP];
            float f = a;
            for (int i = 0; i < 16; ++i) f = __builtin_fmaf(f, 0.99f, c);
            r[0] = __float_as_uint(f);
            r[1] = __float_as_uint(a / b);
            r[2] = __float_as_uint(sqrtf(fabsf(a) + 0.5f));
            r[3] = __float_as_uint(sinf(a));
            r[4] = __float_as_uint(cosf(c));
            r[5] = __float_as_uint(atan2f(a, b));
            r[6] = __float_as_uint(expf(c));
            r[7] = __float_as_uint(erff(a));
            unsigned u = __float_as_uint(a);
            for (int i = 0; i < 8; ++i) u = mix(u, i);
            r[8] = u;
            r[9] = __float_as_uint(__builtin_amdgcn_rcpf(b));
            r[10] = __float_as_uint(__builtin_amdgcn_sinf(a * 0.15915494f));
            {   // the packed-fp32 VALU forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): two fp32 operations per lane and instruction
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 p{a, b}, q{c, a}, w{b, c};
                asm volatile("" : "+v"(p), "+v"(q), "+v"(w));
                f32x2 t = p;
                for (int i = 0; i < 8; ++i) t = __builtin_elementwise_fma(t, q, w);
                r[11] = __float_as_uint(t[0]) ^ __float_as_uint(t[1]);
                f32x2 m = p * q; asm volatile("" : "+v"(m)); m = m * w; asm volatile("" : "+v"(m)); m = m * p;
                r[12] = __float_as_uint(m[0]) ^ __float_as_uint(m[1]);
                f32x2 d2 = p + q; asm volatile("" : "+v"(d2)); d2 = d2 + w; asm volatile("" : "+v"(d2)); d2 = d2 + p;
                r[13] = __float_as_uint(d2[0]) ^ __float_as_uint(d2[1]);
            }
            if (evals == 0) { for (int k = 0; k < NOP; ++k) ref[k] = r[k]; }
            else { for (int k = 0; k < NOP; ++k) moved[k] += r[k] != ref[k]; }
            ++evals;
        }
        __syncthreads();
    }
    for (int k = 0; k < NOP; ++k) if (moved[k]) atomicAdd(report + 8 + k, moved[k]);
    if (tid == 0) atomicAdd(report + 6, evals);
}

// Minimal aggressors: which ingredient of the packed-W GEMMs does it take?  kind bits: 1 = 16-bit MFMAs (v_mfma_f32_32x32x16_f16) back to back,
// 2 = 16-byte buffer loads into the registers the MFMAs read (the packed kernels' W path), 4 = LDS-DMA pieces (their A path), 8 = fp32 MFMAs instead.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256, 2) void aggressor_kernel(float* sink, const unsigned* pattern, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(pattern), 0, 0xffffffff, 0x00020000);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    h16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)(0.01f * (lane + e + i)); b[i][e] = (_Float16)(0.02f * (lane - e + i)); }
    for (int it = 0; it < iters; ++it) {
        const int so = ((it + blockIdx.x) % 48) * 1024;
        if constexpr (KIND & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + i * 4096, so, 0));
        }
        if constexpr (KIND & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (wave * 4 + i) * 256), 16, lane * 16, so + i * 1024, 0, 0);
        }
        if constexpr (KIND & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(i + r) & 3], a[i], acc[i], 0, 0, 0);
        }
        if constexpr (KIND & 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)b[(i + r) & 3][0], (float)a[i][0], acc[i], 0, 0, 0);
        }
        if constexpr (KIND & 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
    for (int i = 0; i < 4; ++i) t += (float)b[i][0];
    if (t == 12345.678f) sink[0] = t + (float)smem[tid];
#endif
}

extern "C" __attribute__((visibility("default"))) int aggressor_launch(int kind, float* sink, const unsigned* pattern, int nwg, int iters, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
#define AGG(K) case K: hipLaunchKernelGGL(aggressor_kernel<K>, dim3(nwg), dim3(256), 16384, st, sink, pattern, iters); break;
    switch (kind) { AGG(1) AGG(2) AGG(3) AGG(4) AGG(5) AGG(6) AGG(7) AGG(8) AGG(10) AGG(12) default: return 3; }
#undef AGG
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" __attribute__((visibility("default"))) int canary_ops_launch(unsigned* report, const float* inputs, int nwg, int spin_us, void* stream) {
    hipLaunchKernelGGL(canary_ops_kernel, dim3(nwg), dim3(256), 0, static_cast<hipStream_t>(stream), report, inputs, spin_us);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" __attribute__((visibility("default"))) int canary_trans_launch(unsigned* report, const float* inputs, int nwg, int spin_us, void* stream) {
    hipLaunchKernelGGL(canary_trans_kernel, dim3(nwg), dim3(256), 0, static_cast<hipStream_t>(stream), report, inputs, spin_us);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" __attribute__((visibility("default"))) int canary_launch(unsigned* report, unsigned* gbuf, const unsigned* pattern, int nwg, int lds_bytes, int spin_us, void* stream) {
    static int attr = 0;
    if (attr < lds_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 1;
        attr = lds_bytes;
    }
    hipLaunchKernelGGL(canary_kernel, dim3(nwg), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), report, gbuf, pattern, lds_bytes / 4, spin_us);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
