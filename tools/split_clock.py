"""Shader clock inside the K loop of the packed fp32-split GEMM (s_memtime / s_memrealtime, mmdm_diag_set "split_timeline") as a function of how many CUs work and of
the operand DATA: random operands vs all-zero operands at the same shape -- the evidence that the loop runs against the chip's power management, not against a latency.
usage: python tools/split_clock.py   (GPU)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, math, ctypes as C
from mixermdm_amd import ops, load_library
lib = load_library(); d = torch.device("cuda:0")
vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.mmdm_diag_set(b"split_cfg", 6)
for M, N, K, zero in [(19200, 3072, 1024, False), (19200, 3072, 1024, True), (2048, 3072, 1024, False), (1024, 3072, 1024, False), (512, 3072, 1024, False), (128, 3072, 1024, False)]:
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / math.sqrt(K); b = torch.randn(N, device=d)
    if zero: x.zero_(); w.zero_()
    out = torch.empty(M, N, device=d)
    xs, ws = ops.split_f32(x), ops.split_f32(w)
    wp = ops.split_pack_weight(ws)
    call = lambda: lib.mmdm_linear_split_packed(vp(xs), K, M * K, vp(wp), N * K, vp(b), vp(out), N, 0, 0, M, N, K, 0, None, 0, 0, st())
    for _ in range(300): call()
    nwg = ((M + 127) // 128) * (N // 128)
    tl = torch.zeros(nwg * 4 * 8, device=d, dtype=torch.int64)
    lib.mmdm_diag_set(b"split_timeline", tl.data_ptr()); assert call() == 0; torch.cuda.synchronize(); lib.mmdm_diag_set(b"split_timeline", 0)
    t = tl.view(nwg, 4, 8).double().cpu()
    nkt = t[0, 0, 5].item()
    per = (t[:, :, :5] / nkt).sum(-1).mean().item()
    mhz = (t[:, :, 7] / t[:, :, 6].clamp(min=1)).median().item() * 100
    print(f"M={M:6d} tiles {nwg:5d} {'zero operands' if zero else 'random      '}: {mhz:.0f} MHz in the loop, {per:.0f} clocks per K step")
