"""GPU: the guards of the packed-fp32 sites that remain in the library ("packed-fp32 sites"; LAB_NOTES.md round 5 / 6).

On gfx950 dense VALU code with packed-fp32 instructions in it was seen to compute other bits while its wave shares a SIMD with the packed-W
GEMM kernels (gemm_splitw / gemm_bf16w); the root cause is open.  By construction no kernel of a precision 1-3 handle and no geometry kernel holds
such an instruction (mixermdm_amd/build.py; tests/test_abi_cpu.py disassembles the objects).  What still does -- rowops.o's AdaLN (precision 0 handles,
the stateless entry points) and the fp32 attention kernel's softmax -- can meet ANOTHER handle's packed-W GEMMs on the device; this file holds them
bit-stable beside those aggressors at the step's real shapes (round 5's tools/adaln_victim.py as a test: 0 of 6000 launches moved there)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aggressors():
    from mixermdm_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 8192, 1024, 1024
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).cuda()
    b = torch.randn(N, generator=g).cuda()
    xs, wsp = ops.split_f32(x), ops.split_pack_weight(ops.split_f32(w))
    xb, wbp = ops.to_bf16(x), ops.pack_weight_frag(ops.to_bf16(w))
    xq, xsc = ops.quantize_rows_fp8(x)
    wq, wsc = ops.quantize_rows_fp8(w)
    wqp = ops.pack_weight_frag(wq)
    return {"the packed split GEMM": (lambda: ops.linear_split(xs, wsp, b, packed=True), 40),
            "the packed bf16 GEMM": (lambda: ops.linear_bf16(xb, wbp, b, packed=True), 50),
            "the packed fp8 GEMM": (lambda: ops.linear_fp8(xq, xsc, wqp, wsc, b, "bias", None, out_dtype=torch.bfloat16, packed=True), 60)}


def _moved(f, aggr, n_aggr, rounds=6, per_round=25):
    aux, side = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        ref = f().clone()
    torch.cuda.synchronize()
    moved = launches = overlapped = 0
    for _ in range(rounds):
        with torch.cuda.stream(aux):
            for _ in range(n_aggr):
                aggr()
        outs = []
        with torch.cuda.stream(side):
            for _ in range(per_round):
                outs.append(f())
        overlapped += int(not aux.query())         # the aggressor was still running when the victim's launches had been queued
        torch.cuda.synchronize()
        moved += sum(int(not torch.equal(o, ref)) for o in outs)
        launches += len(outs)
    return moved, launches, overlapped


@pytest.mark.parametrize("rows,T", [(19200, 300), (1196, 299)])
@pytest.mark.parametrize("form", ["fp32", "fp8"])
def test_adaln_keeps_its_bits_beside_the_packed_gemms(aggressors, rows, T, form):
    from mixermdm_amd import ops
    g = torch.Generator().manual_seed(12)
    nseq = rows // T
    h = torch.randn(rows, 1024, generator=g).cuda()
    ss = (torch.randn(nseq, 2048, generator=g) * 0.3).cuda()
    if form == "fp32":
        f = lambda: ops.adaln(h.view(nseq, T, 1024), ss)
    else:
        def f():
            q, sc = ops.adaln_fp8(h.view(nseq, T, 1024), ss)
            return torch.cat([q.view(torch.uint8).flatten().float(), sc.flatten()])
    for name, (aggr, n) in aggressors.items():
        moved, launches, overlapped = _moved(f, aggr, n)
        assert moved == 0, f"AdaLN ({form} rows, {rows} x 1024) beside {name}: {moved} of {launches} launches moved"
        assert overlapped > 0 or rows < 2000, f"{name} never overlapped the victim: the test did not test anything"


def test_fp32_attention_keeps_its_bits_beside_the_packed_gemms(aggressors):
    from mixermdm_amd import ops
    g = torch.Generator().manual_seed(13)
    qkv = torch.randn(64, 300, 3 * 1024, generator=g).cuda()
    f = lambda: ops.attention(qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:], 8)
    for name, (aggr, n) in aggressors.items():
        moved, launches, overlapped = _moved(f, aggr, n, rounds=4, per_round=20)
        assert moved == 0, f"fp32 attention (64 x 8 heads x 300 x 128) beside {name}: {moved} of {launches} launches moved"
