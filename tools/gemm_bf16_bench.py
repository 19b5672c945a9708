import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, statistics
from mixermdm_amd import ops, load_library
lib = load_library(); d = torch.device("cuda:0")
shapes = [(19200,3072,1024,"qkv","bias"),(19200,1024,1024,"out","resid"),(19200,2048,1024,"ffn1","gelu"),(19200,1024,2048,"ffn2","resid"),(19200,512,512,"m.out","resid"),(19200,1536,512,"m.qkv","bias"),(19200,1024,512,"m.ffn1","gelu"),(19200,512,1024,"m.ffn2","resid"),(8192,8192,8192,"sq8k","bias")]
cfgs = [int(c) for c in os.environ.get("CFGS", "0,1,2,3").split(",")]
for M,N,K,name,epi in shapes:
    x = torch.randn(M,K,device=d); w = torch.randn(N,K,device=d)/math.sqrt(K); b = torch.randn(N,device=d)
    xb, wb = ops.to_bf16(x), ops.to_bf16(w)
    extra = torch.randn(M,N,device=d) if epi=="resid" else None
    # correctness vs fp32 matmul of the rounded operands
    ref = torch.nn.functional.linear(xb.float(), wb.float(), b)
    if epi=="gelu": ref = torch.nn.functional.gelu(ref)
    if epi=="resid": ref = ref + extra
    line=f"{name:6s} {M}x{N}x{K} {epi:5s}"
    for c in cfgs:
        lib.mmdm_diag_set(b"bf16_cfg", c)
        out = ops.linear_bf16(xb,wb,b,epi,extra)
        err=(out-ref).abs().max().item()
        res=[]
        for r in range(5):
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(4): ops.linear_bf16(xb,wb,b,epi,extra)
            e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
        ms=statistics.median(res); line+=f" | cfg{c}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:7.1f}TF err {err:.1e}"
    # weights in fragment order, W straight from global memory (gemm_bf16w_kernel): bitwise the plane kernel's result
    if N % 256 == 0 and K % 128 == 0:
        import ctypes as C
        vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        wp = torch.empty_like(wb)
        assert lib.mmdm_pack_weight_frag(vp(wb), 2 * K, vp(wp), N, 2 * K, st) == 0, lib.mmdm_last_error()
        outp = torch.empty(M, N, device=d)
        ex = extra
        call = lambda: lib.mmdm_linear_bf16_packed(vp(xb), K, vp(wp), vp(b), vp(outp), N, 0, M, N, K, ops.EPI[epi], vp(ex), N if ex is not None else 0, 0, st)
        assert call() == 0, lib.mmdm_last_error()
        torch.cuda.synchronize()
        lib.mmdm_diag_set(b"bf16_cfg", -1)
        ref2 = ops.linear_bf16(xb, wb, b, epi, extra)
        for pc, tag in ((12, "128x256"), (11, "128x128")):          # forced packed tile shapes (mmdm_diag_set "bf16_cfg" 12 / 11)
            lib.mmdm_diag_set(b"bf16_cfg", pc)
            assert call() == 0, lib.mmdm_last_error()
            torch.cuda.synchronize()
            same = torch.equal(outp, ref2)
            res=[]
            for r in range(5):
                e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
                for _ in range(4): call()
                e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1)/4)
            ms=statistics.median(res); line+=f" | packed {tag}: {ms*1e3:7.1f}us {2*M*N*K/ms/1e9:7.1f}TF {'==' if same else '!='}"
        lib.mmdm_diag_set(b"bf16_cfg", -1)
    print(line, flush=True)
