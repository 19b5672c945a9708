"""AdaLN by linearity vs the stand-alone pass, per layer shape at M = nseq x 300 (GPU box; NSEQ=64 -> M = 19 200, NSEQ=4 -> the B = 1 shapes).
   consumer: (a) adaln kernel + plain GEMM   (b) lnfold GEMM on the scaled copy   (c) plain GEMM alone
   producer: residual GEMM plain / with statistics + one scaled copy / + two copies"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops
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
nseq, T = int(os.environ.get("NSEQ", "64")), 300
M = nseq * T
for K, N, epi, name in [(1024, 3072, "bias", "qkv"), (1024, 2048, "gelu", "ffn1"), (1024, 2048, "bias", "ca kv"), (1024, 1024, "bias", "ca q"), (512, 1536, "bias", "m.qkv"), (512, 1024, "gelu", "m.ffn1")]:
    if os.environ.get("ONLY") and name not in os.environ["ONLY"].split(","): continue
    h = torch.randn(M, K, device=d) * 2 + 0.3
    w0 = torch.randn(K, K, device=d) / math.sqrt(K); b0 = torch.randn(K, device=d)
    x = torch.randn(M, K, device=d)
    ss = torch.randn(nseq, 2 * K, device=d) * 0.3
    hh, stats, hs1, _ = ops.linear_scaled(x, w0, b0, "resid", h, ss, T)
    w1 = torch.randn(N, K, device=d) / math.sqrt(K); b1 = torch.randn(N, device=d)
    uc = torch.cat([(1 + ss[:, :K]) @ w1.T, ss[:, K:] @ w1.T + b1], dim=1).contiguous()
    ta = t(lambda: ops.linear(ops.adaln(hh.view(nseq, T, K), ss).view(M, K), w1, b1, epi))
    tb = t(lambda: ops.linear_lnfold(hs1, stats, uc, T, w1, epi))
    xn = ops.adaln(hh.view(nseq, T, K), ss).view(M, K)
    tc = t(lambda: ops.linear(xn, w1, b1, epi))
    tp0 = t(lambda: ops.linear(x, w0, b0, "resid", h))
    tp1 = t(lambda: ops.linear_scaled(x, w0, b0, "resid", h, ss, T))
    tp2 = t(lambda: ops.linear_scaled(x, w0, b0, "resid", h, ss, T, scale2=ss))
    from mixermdm_amd._lib import diag
    diag("gemm_tst", 0)             # direct (row-per-lane) stores instead of the LDS transposition
    td0 = t(lambda: ops.linear(x, w0, b0, "resid", h)); td1 = t(lambda: ops.linear_scaled(x, w0, b0, "resid", h, ss, T)); td2 = t(lambda: ops.linear_scaled(x, w0, b0, "resid", h, ss, T, scale2=ss))
    diag("gemm_tst", 1)
    print(f"{name:7s} {M}x{N}x{K} {epi:5s}: adaln+gemm {ta:7.1f} us | lnfold {tb:7.1f} us | gemm alone {tc:7.1f} us  (adaln pass {ta - tc:5.1f}, the fold costs the GEMM {tb - tc:+6.1f}) "
          f"| producer {K}x{K} resid: {tp0:6.1f} -> stats + 1 copy {tp1:6.1f} -> + 2 copies {tp2:6.1f} us (direct stores: {td0:6.1f} / {td1:6.1f} / {td2:6.1f})", flush=True)
