// Is an fp32 MFMA accumulator the SAME k-ordered fp32 chain whatever the instruction shape?  C = bias + A W^T for one 32 x 32 output block, K = 1024 and 2048, with
//   (a) v_mfma_f32_32x32x2_f32 (the GEMM's instruction: 2 k per MFMA), (b) v_mfma_f32_16x16x4_f32 (4 k per MFMA, four 16 x 16 blocks), (c) a scalar fmaf chain in k order --
// compared bit for bit.  (MI355X_MICROARCH.md says the f32-input MFMAs are "exact f32 (= fmaf chain, bitwise)"; this pins it on the box for the question whether a
// small-M GEMM on 16 x 16 blocks -- four times as many, shorter-latency accumulator chains -- would keep every bit of the 32 x 32 form.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_bits.hip -o /tmp/mcb && /tmp/mcb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [32][K], W [32][K] row-major; C[m][n] = bias[n] + sum_k A[m][k] W[n][k]
__global__ void k32(const float* A, const float* W, const float* bias, float* C, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    // D^T = W A^T as in gemm_f32.hip is one choice; here plainly D = A W^T: a = A[m = l31][k = lh], b = W[n = l31][k = lh]; D[m][n]: lane (n = l31, lh) holds rows m = 8 i + 4 ... (32x32 map)
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = bias[l31];
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + lh], W[l31 * K + k + lh], acc, 0, 0, 0);
    // 32x32 accumulator map: element e of lane (n = l31, lh): row m = 8 (e / 4) + 4 lh + (e % 4)
    for (int e = 0; e < 16; ++e) C[(8 * (e / 4) + 4 * lh + (e % 4)) * 32 + l31] = acc[e];
}
__global__ void k16(const float* A, const float* W, const float* bias, float* C, int K) {
    const int lane = threadIdx.x, l15 = lane & 15, g = lane >> 4;
    for (int bm = 0; bm < 2; ++bm)
        for (int bn = 0; bn < 2; ++bn) {
            f32x4 acc;
            for (int e = 0; e < 4; ++e) acc[e] = bias[16 * bn + l15];
            for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * bm + l15) * K + k + g], W[(16 * bn + l15) * K + k + g], acc, 0, 0, 0);
            for (int e = 0; e < 4; ++e) C[(16 * bm + 4 * g + e) * 32 + 16 * bn + l15] = acc[e];      // 16x16 map: lane (n = l15, g) holds rows 4 g + e
        }
}
__global__ void kf(const float* A, const float* W, const float* bias, float* C, int K) {
    const int m = threadIdx.x >> 5, n = threadIdx.x & 31;
    float acc = bias[n];
    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(A[m * K + k], W[n * K + k], acc);
    C[m * 32 + n] = acc;
}
int main() {
    for (int K : {1024, 2048}) {
        float *hA = (float*)malloc(32 * K * 4), *hW = (float*)malloc(32 * K * 4), hb[32];
        srand(K);
        for (int i = 0; i < 32 * K; ++i) { hA[i] = (float)rand() / (float)RAND_MAX * 2.f - 1.f; hW[i] = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 0.05f; }
        for (int i = 0; i < 32; ++i) hb[i] = (float)rand() / (float)RAND_MAX - 0.5f;
        float *A, *W, *b, *C; (void)hipMalloc(&A, 32 * K * 4); (void)hipMalloc(&W, 32 * K * 4); (void)hipMalloc(&b, 128); (void)hipMalloc(&C, 3 * 1024 * 4);
        (void)hipMemcpy(A, hA, 32 * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, hW, 32 * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 128, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, A, W, b, C, K);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, A, W, b, C + 1024, K);
        hipLaunchKernelGGL(kf, dim3(1), dim3(1024), 0, 0, A, W, b, C + 2048, K);
        float h[3 * 1024]; (void)hipMemcpy(h, C, sizeof(h), hipMemcpyDeviceToHost);
        int d16 = 0, dff = 0; double ref_err = 0;
        for (int i = 0; i < 1024; ++i) {
            d16 += memcmp(&h[i], &h[1024 + i], 4) != 0; dff += memcmp(&h[i], &h[2048 + i], 4) != 0;
            double r = hb[i & 31]; for (int k = 0; k < K; ++k) r += (double)hA[(i >> 5) * K + k] * hW[(i & 31) * K + k];
            if (fabs(h[i] - r) > ref_err) ref_err = fabs(h[i] - r);
        }
        printf("K = %d: 32x32x2 vs 16x16x4: %d of 1024 elements differ; 32x32x2 vs fmaf chain: %d differ; max |32x32x2 - float64| %.2e\n", K, d16, dff, ref_err);
        (void)hipFree(A); (void)hipFree(W); (void)hipFree(b); (void)hipFree(C); free(hA); free(hW);
    }
    return 0;
}
