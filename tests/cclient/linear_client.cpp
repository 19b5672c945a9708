// Native client of the C ABI (no Python, no torch): one mmdm_linear_f32 call with a GELU epilogue on device buffers it allocates itself,
// checked against a host loop.  Built by tests/test_gpu_cclient.py with hipcc; prints "OK <max abs err>" or "FAIL ...".
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "mmdm.h"

int main() {
    const int M = 300, N = 96, K = 64;
    std::vector<float> A(M * K), W(N * K), b(N), C(M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : A) v = rnd();
    for (auto& v : W) v = rnd() * 0.2f;
    for (auto& v : b) v = rnd();
    float *dA, *dW, *db, *dC;
    if (hipMalloc(&dA, A.size() * 4) || hipMalloc(&dW, W.size() * 4) || hipMalloc(&db, b.size() * 4) || hipMalloc(&dC, C.size() * 4)) { printf("FAIL hipMalloc\n"); return 1; }
    hipStream_t st;
    if (hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice) || hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice) ||
        hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice) || hipStreamCreate(&st)) { printf("FAIL copy / stream\n"); return 1; }
    int rc = mmdm_linear_f32(dA, K, dW, K, db, dC, N, M, N, K, MMDM_EPI_BIAS_GELU, nullptr, 0, 0, st);
    if (rc) { printf("FAIL rc=%d %s\n", rc, mmdm_last_error()); return 1; }
    if (hipStreamSynchronize(st) || hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost)) { printf("FAIL sync / copy back\n"); return 1; }
    double worst = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double z = b[n];
            for (int k = 0; k < K; ++k) z += (double)A[m * K + k] * W[n * K + k];
            const double ref = 0.5 * z * (1.0 + erf(z * 0.70710678118654752440));
            worst = fmax(worst, fabs(ref - C[m * N + n]));
        }
    rc = mmdm_linear_f32(nullptr, K, dW, K, db, dC, N, M, N, K, MMDM_EPI_BIAS, nullptr, 0, 0, st);      // error path: status code + message
    if (rc == 0 || !mmdm_last_error()[0]) { printf("FAIL no error for a null operand\n"); return 1; }
    printf(worst < 1e-5 ? "OK %.3e\n" : "FAIL %.3e\n", worst);
    return worst < 1e-5 ? 0 : 1;
}
