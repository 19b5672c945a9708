"""The residual GEMM that also writes the next block's AdaLN (mmdm_linear_f32_ln) against the plain residual GEMM + the stand-alone pass,
per layer shape at M = 19 200, T = 300 (GPU box; warm clocks)."""
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
nseq, T = 64, 300
M = nseq * T
for N, K, name in [(1024, 1024, "out-proj"), (1024, 2048, "ffn-2"), (512, 512, "m.out-proj"), (512, 1024, "m.ffn-2")]:
    x = torch.randn(M, K, device=d); h = torch.randn(M, N, device=d) * 2 + 0.3
    w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
    ss = torch.randn(nseq, 2 * N, device=d) * 0.3
    ta = t(lambda: ops.linear(x, w, b, "resid", h))
    y = ops.linear(x, w, b, "resid", h)
    tb = t(lambda: ops.adaln(y.view(nseq, T, N), ss))
    tab = t(lambda: ops.adaln(ops.linear(x, w, b, "resid", h).view(nseq, T, N), ss))
    _, _, work = ops.linear_ln(x, w, b, "resid", h, ss, T)
    tc = t(lambda: ops.linear_ln(x, w, b, "resid", h, ss, T, work=work))
    print(f"{name:10s} {M}x{N}x{K}: residual GEMM {ta:7.1f} us | AdaLN pass {tb:5.1f} us | both in sequence {tab:7.1f} us | GEMM writing the AdaLN too {tc:7.1f} us ({tc - ta:+6.1f} vs the plain GEMM)", flush=True)
