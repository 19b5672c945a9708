"""Packed bf16 GEMM (gemm_bf16w_kernel): how much of a workgroup's life is its K loop, and the shader clock inside it
(s_memtime / s_memrealtime; mmdm_diag_set "bf16_timeline").  the matrix pipe never idles at 32 (128 x 256 tile) or 16 (128 x 128) MFMAs x 32 cycles per wave and step, two waves per SIMD."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, ctypes as C
from mixermdm_amd import ops, load_library
lib = load_library(); d = torch.device("cuda:0")
vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
FP8 = os.environ.get("FP8") == "1"          # FP8=1: the fp8-operand instantiations (QKV -> bf16, FFN-2 + residual -> fp32, ...)
for M, N, K, epi in [(19200, 3072, 1024, "bias"), (19200, 1024, 1024, "resid"), (19200, 1024, 2048, "resid"), (8192, 8192, 8192, "bias")]:
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
    extra = torch.randn(M, N, device=d) if epi == "resid" else None
    if FP8:
        xq, xs = ops.quantize_rows_fp8(x); wq, ws = ops.quantize_rows_fp8(w); wp = ops.pack_weight_frag(wq)
        call = lambda: ops.linear_fp8(xq, xs, wp, ws, b, epi, extra, out_dtype=torch.float32 if epi == "resid" else torch.bfloat16, packed=True)
    else:
        xb, wb = ops.to_bf16(x), ops.to_bf16(w); wp = ops.pack_weight_frag(wb)
        call = lambda: ops.linear_bf16(xb, wp, b, epi, extra, packed=True)
    for _ in range(int(os.environ.get("WARM", "400"))): call()
    kern = lib.mmdm_last_gemm_kernel().decode()
    bn = 256 if kern.endswith("42>") else 128
    nwg = ((M + 127) // 128) * (N // bn)
    tl = torch.zeros(nwg * 4, device=d, dtype=torch.int64)
    lib.mmdm_diag_set(b"bf16_timeline", tl.data_ptr()); call(); torch.cuda.synchronize(); lib.mmdm_diag_set(b"bf16_timeline", 0)
    raw1 = tl.view(nwg, 4)[:, 1].cpu()
    iss = (raw1 >> 32).double()                     # ticks from the end of the K loop until the epilogue's last instruction has issued
    tl.view(nwg, 4)[:, 1] &= 0xffffffff
    t = tl.view(nwg, 4).double().cpu()
    mhz = (100.0 * t[:, 0] / t[:, 1].clamp(min=1)).median().item()
    raw3 = tl.view(nwg, 4)[:, 3].cpu()
    nkt = float(raw3[0].item() & 0xffffffff)
    pro = (raw3 >> 32).double()                     # ticks from kernel entry to the K loop
    t[:, 3] = nkt
    per_step = (t[:, 0] / nkt).median().item()
    mf = (2 if "fp8" in kern else 1) * 4 * (bn // 128) * 4          # MFMAs per wave and step
    print(f"{M}x{N}x{K} {epi:5s} {kern}: loop {t[:, 1].median().item() / 100:.1f} us of a {t[:, 2].median().item() / 100:.1f} us workgroup "
          f"({100 * (t[:, 1] / t[:, 2]).median().item():.0f} %), {per_step:.0f} shader clocks per K step (matrix pipe never idle: {mf * 32 * 2}), clock in the loop {mhz:.0f} MHz; before the loop {pro.median().item() / 100:.1f} us, after it {(t[:, 2] - t[:, 1] - pro).median().item() / 100:.1f} us (epilogue issued after {iss.median().item() / 100:.1f} us, then its stores drain)")
