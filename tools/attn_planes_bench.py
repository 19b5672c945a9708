"""Attention kernels at the layer shapes: the fp32 MFMA kernel, the plane forms (Q K^T on the 16-bit cores, fp32 P.V), the all-bf16 form and the
fp32-split mode's fp16 two-plane form.   usage: python tools/attn_planes_bench.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd import ops, load_library
lib = load_library()
d = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for nseq, T, H, dh in [(64, 300, 8, 128), (128, 300, 8, 64), (256, 300, 8, 128), (4, 299, 8, 128)]:
    qkv = torch.randn(nseq, T, 3 * H * dh, device=d)
    HD = H * dh
    q, k, v = qkv[..., :HD], qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    qp3, kp3 = ops.bf16_split3(q.contiguous()), ops.bf16_split3(k.contiguous())
    qp1, kp1 = q.contiguous().bfloat16()[None].contiguous(), k.contiguous().bfloat16()[None].contiguous()
    qb = qkv.bfloat16()
    qs = ops.split_f32(qkv)
    fl = 4.0 * nseq * H * T * (T + 1) * dh
    rows = [("fp32 16x16x4", lambda: ops.attention(q, k, v, H), 0), ("QK^T 3 planes", lambda: ops.attention_planes(qp3, kp3, v, H), 0),
            ("QK^T 1 plane", lambda: ops.attention_planes(qp1, kp1, v, H), 0)]
    def bf16_16():
        lib.mmdm_diag_set(b"attn_kc32", 0)
        try: return ops.attention_bf16(qb[..., :HD], qb[..., HD:2 * HD], qb[..., 2 * HD:], H)
        finally: lib.mmdm_diag_set(b"attn_kc32", 1)
    rows.append(("all-bf16, 16-key chunks", bf16_16, 0))
    rows.append(("all-bf16, 32-key chunks", lambda: ops.attention_bf16(qb[..., :HD], qb[..., HD:2 * HD], qb[..., 2 * HD:], H), 0))
    rows.append(("fp16 split", lambda: ops.attention_split(qs[..., :HD], qs[..., HD:2 * HD], qs[..., 2 * HD:], H), 0))
    for name, fn, _ in rows:
        ms = timeit(fn)
        print(f"nseq={nseq:3d} dh={dh:3d} {name:22s} {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF(alg)", flush=True)
