"""CPU experiment behind the operand format of the fp32-split mode (csrc/gemm_split.hip, kernels.h mmdm_split2): error against a float64 product of
  sgemm      torch's CPU fp32 GEMM,
  chain16    fp32 accumulation in k-ordered blocks of 16 (the shape of an MFMA accumulation chain),
  bf16x3     the exact three-way bf16 split of rounds 1-3 (six products per block, one accumulator),
  fp16x2     the two-way fp16 split of round 4 (three products per block, hi / lo accumulators, lo carried at 2^11),
for normal, heavy-tailed, post-GELU and tiny operands.  fp16 x fp16 and bf16 x bf16 products are exact in fp32, so float32 matmuls of the
rounded planes reproduce the MFMA arithmetic up to the order inside a 16-block.   usage: python tools/split_numerics.py   (CPU, ~1 min)"""
import numpy as np, torch
torch.manual_seed(0)
M,K,N = 2048,1024,1024
def run(A, W, tag):
    ref = (A.double() @ W.double().T)
    # fp32 k-ordered-ish: torch CPU sgemm
    c32 = (A @ W.T).double()
    # k-ordered chains of 16 (approximate the MFMA chain): chunked accumulate
    acc = torch.zeros(M,N)
    for k in range(0,K,16): acc = acc + (A[:,k:k+16] @ W[:,k:k+16].T)
    cch = acc.double()
    # bf16x3
    def split3(x):
        x1 = x.bfloat16().float(); r = x-x1; x2 = r.bfloat16().float(); r2 = r-x2; x3 = r2.bfloat16().float(); return x1,x2,x3
    a1,a2,a3 = split3(A); w1,w2,w3 = split3(W)
    acc = torch.zeros(M,N)
    for k in range(0,K,16):
        s = slice(k,k+16)
        acc = acc + a3[:,s]@w1[:,s].T; acc = acc + a2[:,s]@w2[:,s].T; acc = acc + a1[:,s]@w3[:,s].T
        acc = acc + a2[:,s]@w1[:,s].T; acc = acc + a1[:,s]@w2[:,s].T; acc = acc + a1[:,s]@w1[:,s].T
    cb3 = acc.double()
    # fp16x2, second piece scaled 2^11, two accumulators
    S = 2048.0
    def split2(x):
        x1 = x.half().float(); x2 = ((x-x1)*S).half().float(); return x1,x2
    a1,a2 = split2(A); w1,w2 = split2(W)
    hi = torch.zeros(M,N); lo = torch.zeros(M,N)
    for k in range(0,K,16):
        s = slice(k,k+16)
        hi = hi + a1[:,s]@w1[:,s].T
        lo = lo + a1[:,s]@w2[:,s].T; lo = lo + a2[:,s]@w1[:,s].T
    c2 = (hi + lo/S).double()
    # representational error alone
    c2x = ((a1.double()+a2.double()/S) @ (w1.double()+w2.double()/S).T)
    sc = ref.abs().mean()
    for nm,c in [("sgemm",c32),("chain16",cch),("bf16x3",cb3),("fp16x2",c2),("fp16x2 operands only",c2x)]:
        e=(c-ref).abs()
        print(f"{tag:10s} {nm:22s} mean {e.mean()/sc:.3e}  p99.9 {e.flatten().kthvalue(int(e.numel()*0.999)).values/sc:.3e} max {e.max()/sc:.3e}")
A = torch.randn(M,K)*1.5; W = torch.randn(N,K)*0.02
run(A,W,"normal")
A = torch.randn(M,K)*torch.exp(torch.randn(M,K)*2); W = torch.randn(N,K)*0.02*torch.exp(torch.randn(N,K)*1.5)
run(A,W,"heavytail")
A = torch.nn.functional.gelu(torch.randn(M,K)*2); run(A, torch.randn(N,K)*0.02, "gelu")
A = torch.randn(M,K)*1e-3; W=torch.randn(N,K)*1e-3
run(A,W,"tiny")
