"""AdaLN fused into the fp32 GEMM vs the stand-alone pass, per layer shape at M = 19 200, T = 300 (GPU box).
   (a) adaln kernel + plain GEMM   (b) fused consumer GEMM   (c) plain GEMM alone   (d) residual GEMM with / without the statistics output"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops
from mixermdm_amd._lib import load_library
d = torch.device("cuda:0")
_w = torch.randn(4096, 4096, device=d)
for _ in range(60): ops.linear(_w, _w)
def t(f, reps=4, rounds=7):
    f(); res = []
    for _ in range(rounds):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1) / reps * 1e3)
    return statistics.median(res)
nseq, T = 64, 300
M = nseq * T
for K, N, epi, name in [(1024, 3072, "bias", "qkv"), (1024, 2048, "gelu", "ffn1"), (1024, 2048, "bias", "ca kv"), (1024, 1024, "bias", "ca q"), (512, 1536, "bias", "m.qkv"), (512, 1024, "gelu", "m.ffn1")]:
    if os.environ.get("ONLY") and name not in os.environ["ONLY"].split(","): continue
    h = torch.randn(M, K, device=d) * 2 + 0.3
    w0 = torch.randn(K, K, device=d) / math.sqrt(K); b0 = torch.randn(K, device=d)
    x = torch.randn(M, K, device=d)
    hh, stats = ops.linear_stats(x, w0, b0, "resid", h)
    w1 = torch.randn(N, K, device=d) / math.sqrt(K); b1 = torch.randn(N, device=d)
    ss = torch.randn(nseq, 2 * K, device=d) * 0.3
    ta = t(lambda: ops.linear(ops.adaln(hh.view(nseq, T, K), ss).view(M, K), w1, b1, epi))
    tb = t(lambda: ops.linear_adaln(hh, stats, ss, T, w1, b1, epi))
    xn = ops.adaln(hh.view(nseq, T, K), ss).view(M, K)
    tc = t(lambda: ops.linear(xn, w1, b1, epi))
    lib = load_library()
    lib.mmdmx_set_gemm_cfg(31); t31 = t(lambda: ops.linear(xn, w1, b1, epi))
    lib.mmdmx_set_gemm_cfg(34); t34 = t(lambda: ops.linear(xn, w1, b1, epi))
    lib.mmdmx_set_gemm_cfg(-1)
    tp0 = t(lambda: ops.linear(x, w0, b0, "resid", h))
    tp1 = t(lambda: ops.linear_stats(x, w0, b0, "resid", h))
    print(f"{name:7s} {M}x{N}x{K} {epi:5s}: adaln+gemm {ta:7.1f} us | fused {tb:7.1f} us | gemm alone {tc:7.1f} us  (adaln pass {ta - tc:5.1f}, fusion costs the GEMM {tb - tc:+6.1f}; 128x128 4 / 5 stages plain {t31:6.1f} / {t34:6.1f}) | producer {K}x{K} resid: {tp0:6.1f} -> with stats {tp1:6.1f} us", flush=True)
