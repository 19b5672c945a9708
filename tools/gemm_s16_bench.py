"""Small-launch fp32 GEMM (gemm_s16_kernel: 16x16x4 chains) against the production 64 x 64-tile launch on the reference's B = 1 call shapes:
bits (must be identical for every epilogue) and microseconds.  usage (GPU box): python tools/gemm_s16_bench.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
CFGS = [int(c) for c in os.environ.get('S16_CFGS', '0,-1,14').split(',')]      # 0 = the 64 x 64 launch, -1 = the automatic dispatch (s16 / mix / production), 13 .. 24 forced forms
shapes = [(1196, 3072, 1024, "bias"), (1196, 1024, 1024, "resid"), (1196, 2048, 1024, "bias"), (1196, 2048, 1024, "gelu"), (1196, 1024, 2048, "resid"),
          (1196, 1536, 512, "bias"), (1196, 512, 512, "resid"), (1196, 1024, 512, "gelu"), (1196, 512, 1024, "resid"),
          (240, 3072, 1024, "bias"), (240, 1024, 1024, "resid"), (240, 1024, 2048, "resid"), (601, 1028, 1024, "silu"), (77, 1024, 96, "quickgelu"),
          (1196, 1024, 1024, "pe"), (6, 1024, 1024, "silu"), (3588, 1024, 1024, "resid"), (1196, 516, 512, "sigmoid")]
_w = torch.randn(4096, 4096, device=d)
for _ in range(40): ops.linear(_w, _w)
torch.cuda.synchronize()
bad = 0
for M, N, K, epi in shapes:
    g = torch.Generator(device=d); g.manual_seed(M * 7 + N)
    x = torch.randn(M, K, device=d, generator=g); w = torch.randn(N, K, device=d, generator=g) / math.sqrt(K); b = torch.randn(N, device=d, generator=g)
    extra = torch.randn(M if epi == "resid" else 299, N, device=d, generator=g) if epi in ("resid", "pe") else None
    period = 299 if epi == "pe" else 0
    outs, line = {}, f"{M:5d}x{N:4d}x{K:4d} {epi:9s}"
    for c in CFGS:
        lib.mmdm_diag_set(b"gemm_s16", c)
        out = torch.full((M, N), float("nan"), device=d)
        ops.linear(x, w, b, epi, extra, period, out=out)
        outs[c] = out.clone()
        ts = []
        for r in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.linear(x, w, b, epi, extra, period, out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        same = torch.equal(outs[c], outs[0]) and bool(torch.isfinite(outs[c]).all())
        bad += 0 if same else 1
        line += f" | {c:2d}: {statistics.median(ts) * 1e3:6.1f}us" + (f" [{lib.mmdm_last_gemm_kernel().decode()}]" if c == -1 else "") + ("" if same else " DIFFERENT BITS")
    ref = (x.double() @ w.double().T + b.double())
    print(line + f" | max |prod - f64 pre-activation| {'-' if epi not in ('bias',) else format(float((outs[0].double() - ref).abs().max()), '.2e')}", flush=True)
lib.mmdm_diag_set(b"gemm_s16", -1)
print("every forced configuration bit-identical to the production launch" if not bad else f"{bad} (shape, configuration) pairs differ")
sys.exit(1 if bad else 0)
