// Persistent packed fp8 GEMM (gemm_fp8p.hip): argument block and launcher, shared with gemm_bf16.hip's dispatch.
#pragma once
#include <hip/hip_runtime.h>

struct Fp8pArgs {
    const void* A;            // [M][lda] fp8 e4m3 rows
    const void* W;            // packed fragments (mmdm_pack_weight_frag of [N][K] fp8)
    const float* a_scale;     // [M] or nullptr (a_const)
    const float* w_scale;     // [N]
    const float* bias;        // [N]
    void* C;                  // [M][ldc] fp32 / bf16 / fp8
    const float* extra;       // residual / PE rows (fp32 output only) or nullptr
    int lda, ldc, ld_extra;
    int M, N, K;
    int epilogue;             // MMDM_EPI_*
    int period;               // PE epilogue: row period
    int out_mode;             // 0 fp32, 1 bf16, 2 fp8 (value * out_scale)
    float a_const, out_scale;
    int mt, nt, ntiles;       // filled by the launcher
    unsigned long long* tl;   // diagnostic stamps (tools/fp8p_timeline.py) or nullptr
};

// true if the persistent kernel covers this call (shape, epilogue, pointers); the caller falls back to gemm_bf16w_kernel otherwise
bool mmdm_fp8p_covers(const Fp8pArgs& a);
int mmdm_fp8p_launch(Fp8pArgs a, hipStream_t st);
int mmdm_fp8p_init(void);
