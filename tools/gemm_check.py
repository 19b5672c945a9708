"""Correctness of a forced GEMM tile configuration against float64 (GPU box): CFG=<n> python tools/gemm_check.py"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
ops.linear(torch.zeros(8, 64, device=d), torch.zeros(8, 64, device=d))     # runs the lazy init (which resets the cfg) first
cfgs = [int(c) for c in os.environ.get("CFGS", "30").split(",")]
g = torch.Generator().manual_seed(0)
for cfg in cfgs:
    lib.mmdm_diag_set(b"gemm_cfg", cfg)
    worst = 0.0
    for M, N, K, epi in [(19200, 1024, 1024, "resid"), (1200, 3072, 1024, "bias"), (300, 2048, 1024, "gelu"), (777, 512, 2048, "resid"), (130, 136, 64, "bias"),
                         (19200, 512, 512, "resid"), (257, 1024, 48, "bias"), (64, 64, 80, "pe")]:
        x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / math.sqrt(K); b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g) if epi in ("resid",) else (torch.randn(30, N, generator=g) if epi == "pe" else None)
        ref = F.linear(x.double(), w.double(), b.double())
        if epi == "gelu": ref = F.gelu(ref)
        if epi == "resid": ref = ref + r.double()
        if epi == "pe": ref = ref + r.double()[torch.arange(M) % 30]
        got = ops.linear(x.to(d), w.to(d), b.to(d), epi, r.to(d) if r is not None else None, period=30 if epi == "pe" else 0)
        kern = lib.mmdm_last_gemm_kernel().decode()
        err = (got.cpu().double() - ref).abs().max().item()
        worst = max(worst, err / math.sqrt(K / 1024))
        print(f"cfg {cfg} {M}x{N}x{K} {epi:5s} {kern:34s} max err {err:.2e}", flush=True)
    a = ops.linear(x.to(d), w.to(d), b.to(d)); bb = ops.linear(x.to(d), w.to(d), b.to(d))
    assert torch.equal(a, bb)
    print(f"cfg {cfg}: worst scaled error {worst:.2e} {'OK' if worst < 3e-5 else 'FAIL'}")
lib.mmdm_diag_set(b"gemm_cfg", -1)
