import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd import ops
d = torch.device("cuda:0")
for nseq, T, H, dh in [(64, 300, 8, 128), (128, 300, 8, 64)]:
    qkv = torch.randn(nseq, T, 3 * H * dh, device=d)
    q, k, v = qkv[..., :H * dh], qkv[..., H * dh:2 * H * dh], qkv[..., 2 * H * dh:]
    qp3, kp3 = ops.bf16_split3(q.contiguous()), ops.bf16_split3(k.contiguous())
    qp1, kp1 = q.contiguous().bfloat16()[None].contiguous(), k.contiguous().bfloat16()[None].contiguous()
    fl = 4.0 * nseq * H * T * (T + 1) * dh
    for name, fn in [("fp32 16x16x4", lambda: ops.attention(q, k, v, H)), ("QK^T 3 planes", lambda: ops.attention_planes(qp3, kp3, v, H)),
                     ("QK^T 1 plane", lambda: ops.attention_planes(qp1, kp1, v, H))]:
        for _ in range(3): fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"nseq={nseq} dh={dh} {name:14s} {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF(alg)", flush=True)
