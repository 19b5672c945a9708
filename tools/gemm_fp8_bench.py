"""fp8 GEMM at M = 19 200: the LDS-staged kernel vs the packed-W kernel (weights in fragment order), fp32 / bf16 / fp8 outputs."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops, load_library
lib = load_library(); d = torch.device("cuda:0")
shapes = [(19200,3072,1024,"qkv","bias",torch.bfloat16),(19200,1024,1024,"ca q","bias",torch.bfloat16),(19200,2048,1024,"ca kv","bias",torch.bfloat16),(19200,2048,1024,"ffn1","gelu",torch.float8_e4m3fn),(19200,1024,2048,"ffn2","resid",torch.float32),(8192,8192,8192,"sq8k","bias",torch.bfloat16)]
# FP8_SHAPES=qkv,ffn2 selects shapes by name prefix; FP8_ITERS=1 (profiler passes: tools/pmc_fp8.sh) skips the warm-up GEMMs and times one repeat
sel = [t.strip() for t in os.environ.get("FP8_SHAPES", "").split(",") if t.strip()]
if sel: shapes = [sh for sh in shapes if any(sh[3].replace(" ", "").startswith(t) for t in sel)]
ITERS = int(os.environ.get("FP8_ITERS", "5"))
MB = int(os.environ.get("FP8_M", "0"))
if MB: shapes = [(MB,) + sh[1:] for sh in shapes]
if os.environ.get("FP8_LDS_PAD"): lib.mmdm_diag_set(b"bf16_lds_pad", int(os.environ["FP8_LDS_PAD"]))      # extra dynamic LDS per packed-W workgroup: 2 instead of 3 workgroups per CU from ~7 KB up
if os.environ.get("FP8P_GRID"): lib.mmdm_diag_set(b"fp8p_grid", int(os.environ["FP8P_GRID"]))            # workgroups of the persistent kernel (512 = two per CU)
if os.environ.get("FP8_CFG"): lib.mmdm_diag_set(b"bf16_cfg", int(os.environ["FP8_CFG"]))                    # 12 / 13: force the wide / the narrow packed tile
lib.mmdm_diag_set(b"fp8p", 0)          # the packed / staged columns below are gemm_bf16w / gemm_bf16 kernels; the persistent kernel has its own column
_w = torch.randn(4096, 4096, device=d)
for _ in range(60 if ITERS > 1 else 2): ops.linear(_w, _w)
for M,N,K,name,epi,od in shapes:
    x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d)
    if os.environ.get("FP8_ZERO"): x.zero_(); w.zero_()          # all-zero operands: the same instruction stream at a fraction of the matrix cores' power (is the kernel power-bound?)
    xq, xs = ops.quantize_rows_fp8(x); wq, ws = ops.quantize_rows_fp8(w); wp = ops.pack_weight_frag(wq)
    extra = torch.randn(M,N,device=d) if epi=="resid" else None
    line = f"{name:5s} {M}x{N}x{K} {epi:5s} -> {str(od)[6:]:14s}"
    ref = {}
    ONLY = os.environ.get("FP8_ONLY", "")          # e.g. "packed-t": one kernel form only (profiler passes)
    for tst in (1, 0):                       # 1: results leave through the workgroup's LDS transposition (whole lines); 0: direct row-per-lane stores
        lib.mmdm_diag_set(b"bf16_tst", tst)
        for tag, wgt, pk in (("staged", wq, False), ("packed", wp, True)):
            if ONLY and ONLY != f"{tag}-{'t' if tst else 'd'}": continue
            f = lambda: ops.linear_fp8(xq, xs, wgt, ws, b, epi, extra, out_dtype=od, packed=pk)
            o = f(); res=[]
            key = tag
            # (residual epilogues: the transposed form adds the residual last, (b + sum) + r, the direct form starts from b + r: same value, other rounding)
            if key in ref and epi != "resid": assert torch.equal(o.view(torch.uint8), ref[key].view(torch.uint8)), f"transposed epilogue changed bits: {name} {tag}"
            else: ref[key] = o
            for r in range(ITERS):
                e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
                for _ in range(4): f()
                e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
            ms=statistics.median(res); line+=f" | {tag}{'-t' if tst else '-d'}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:7.1f}TF"
    lib.mmdm_diag_set(b"bf16_tst", 1)
    # the persistent kernel (gemm_fp8p.hip: the epilogue of tile i under tile i + 1's K loop), where it covers the shape: bitwise the packed kernel
    if not ONLY or ONLY == "persist":
        lib.mmdm_diag_set(b"fp8p", 2)
        try:
            f = lambda: ops.linear_fp8(xq, xs, wp, ws, b, epi, extra, out_dtype=od, packed=True)
            o = f()
            if lib.mmdm_last_gemm_kernel().decode().startswith("gemm_fp8p"):
                if "packed" in ref and epi != "resid" and not os.environ.get("FP8_NOCHECK"): assert torch.equal(o.view(torch.uint8), ref["packed"].view(torch.uint8)), f"persistent kernel changed bits: {name}"
                res = []
                for r in range(ITERS):
                    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
                    for _ in range(4): f()
                    e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
                ms=statistics.median(res); line+=f" | persist: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:7.1f}TF"
        finally:
            lib.mmdm_diag_set(b"fp8p", 0)
    print(line, flush=True)
