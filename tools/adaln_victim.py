"""Are the AdaLN kernel and the fp32 attention kernel (built WITH packed-fp32 instructions) victims of the gfx950 packed-fp32 hazard (LAB_NOTES.md, round 5)?  The kernel at the
step's shape (19 200 rows, D = 1024; and the B = 1 shape) launched over and over on one stream, each result compared bit for bit with the first,
while the packed-W GEMMs -- the aggressors of tools/canary.hip -- run on another stream.  Prints launches whose output moved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd import ops
g = torch.Generator().manual_seed(11)
M_, N_, K_ = 8192, 1024, 1024
x = torch.randn(M_, K_, generator=g).cuda(); w = (torch.randn(N_, K_, generator=g) * 0.03).cuda(); bb = torch.randn(N_, generator=g).cuda()
xs = ops.split_f32(x); wsp = ops.split_pack_weight(ops.split_f32(w))
xb = ops.to_bf16(x); wbp = ops.pack_weight_frag(ops.to_bf16(w))
aux, side = torch.cuda.Stream(), torch.cuda.Stream()
for rows, T in ((19200, 300), (1196, 299)):
    nseq = rows // T
    h = torch.randn(rows, 1024, generator=g).cuda(); ss = (torch.randn(nseq, 2048, generator=g) * 0.3).cuda()
    forms = {"AdaLN fp32 rows": lambda: ops.adaln(h.view(nseq, T, 1024), ss)}
    if hasattr(ops, "adaln_fp8"):
        def f8():
            q, sc = ops.adaln_fp8(h.view(nseq, T, 1024), ss)
            return torch.cat([q.view(torch.uint8).flatten().float(), sc.flatten()])
        forms["AdaLN fp8 rows + scales"] = f8
    if rows == 19200:
        qkv = torch.randn(64, T, 3 * 1024, generator=g).cuda()
        forms["fp32 attention (attn_mfma_kernel: 46 op_sel producers read by the next instruction), 64 x 8 heads x 300 x 128"] = \
            lambda: ops.attention(qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:], 8)
    for what, f in forms.items():
        with torch.cuda.stream(side):
            ref = f().clone()
        torch.cuda.synchronize()
        for aggr, af, n in (("nothing", None, 0), ("the packed split GEMM", lambda: ops.linear_split(xs, wsp, bb, packed=True), 40), ("the packed bf16 GEMM", lambda: ops.linear_bf16(xb, wbp, bb, packed=True), 50)):
            moved = launches = 0
            for rnd in range(20):
                if af is not None:
                    with torch.cuda.stream(aux):
                        for _ in range(n): af()
                outs = []
                with torch.cuda.stream(side):
                    for _ in range(25): outs.append(f())
                busy = af is not None and not aux.query()
                torch.cuda.synchronize()
                moved += sum(int(not torch.equal(o, ref)) for o in outs); launches += len(outs)
            print("%s, %d rows, beside %s: %d of %d launches moved (aggressor still running at the end of the last round: %s)" % (what, rows, aggr, moved, launches, busy), flush=True)
