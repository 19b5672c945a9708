"""Two handles side by side, ONE step at a time -- the hunt for round 5's wrong motions (LAB_NOTES.md: a packed-fp32 hazard on gfx950).

usage: [MODE=...] python tools/overlap_bisect.py [precision of handle A] [precision of handle B]
  MODE=steps (default): every step is an independent trial -- both handles are put back on the sequential run's state of step k-1, one step of
      each is queued back to back (nothing synchronised in between), the five state buffers are compared with the sequential run's and the place
      of the wrong numbers is printed.  PROBE_EAGER=1: eager launches; NSTEP=.
  MODE=aggressor / victim_torch / gemm_aggr: torch kernels beside a handle, a torch matmul as the victim, ONE stand-alone library GEMM as the aggressor.
  MODE=canary: tools/canary.hip (build/libcanary.so; CANARY_LIB= another build of it) beside the steps; CANARY_TRANS=1 the rotation round trip,
      CANARY_OPS=1 single operations, CANARY_CHAIN=1 the first intermediate that moves; CANARY_AGGR=1 adds library GEMMs and torch matmuls as
      aggressors, CANARY_VARIANTS=1 the packed kernels' variants, CANARY_MICRO=1 the micro-aggressors, CU_SPLIT=1 victim and aggressor on
      disjoint halves of the CUs.
  MODE=skip / which (+ POISON=1): need the debug build of tools/mk_debug_lib.py (MMDM_LIB=build/libmmdm_debug.so): kernel classes left out of B's
      step graph; A's allocations that differ, by name, after an overlapped step.
With the shipped library (geometry / row kernels built without packed-fp32 instructions) every mode reports zero."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mixermdm_amd.sampler import Sampler
from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
precB = sys.argv[2] if len(sys.argv) > 2 else prec
G = os.environ.get("PROBE_EAGER") != "1"
NSTEP = int(os.environ.get("NSTEP", "16"))
sd = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS); st = synthetic_stats()
def fresh(prec=prec):
    t = Sampler(d_heads=8, m_heads=8, max_batch=1, max_frames=300, precision=prec, **FULL_DIMS)
    t.load_state_dict(sd); t.set_norm_stats(st["mean_hml"], st["std_hml"], st["mean_ih"], st["std_ih"]); t.prepare(); t.set_schedule("ddim50")
    return t
A, B = fresh(prec), fresh(precB)
ia = tuple(t.cuda() for t in synthetic_inputs(1, 181, seed_cond=181, seed_x=182))
ib = tuple(t.cuda() for t in synthetic_inputs(1, 263, seed_cond=263, seed_x=264))
NAMES = ("x", "x2", "pred_xstart", "pred_xstart2", "model_out")
refA_full = A.sample(*ia, use_graph=G); refB_full = B.sample(*ib, use_graph=G)

def where(o, r):
    m = (o != r).reshape(-1, o.shape[-1])
    rows = m.any(1).nonzero().flatten(); cols = m.any(0).nonzero().flatten()
    return "%d el, rows %d in [%d, %d], cols %d in [%d, %d], max|d| %.2e" % (int(m.sum()), rows.numel(), int(rows.min()), int(rows.max()), cols.numel(),
                                                                                int(cols.min()), int(cols.max()), (o - r).abs().max().item())

if os.environ.get("MODE", "steps") == "steps":
    def seq(h, inp):
        h.begin(*inp); out = []
        s0 = h.state(); out.append({k: s0[k].clone() for k in ("x", "x2")})
        for k in range(NSTEP):
            h.run(1, G); s = h.state()
            out.append({n: s[n].clone() for n in NAMES})
        return out
    ra, rb = seq(A, ia), seq(B, ib)
    ra2 = seq(A, ia)
    print(prec, "graph" if G else "eager", "sequential run reproducible:", all(torch.equal(ra[k][n], ra2[k][n]) for k in range(1, NSTEP + 1) for n in NAMES), flush=True)
    A.begin(*ia); B.begin(*ib); A.synchronize(); B.synchronize()
    nbad = 0
    for k in range(1, NSTEP + 1):
        for h, r in ((A, ra), (B, rb)):
            s = h.state()
            s["x"].copy_(r[k - 1]["x"]); s["x2"].copy_(r[k - 1]["x2"])
        torch.cuda.synchronize()
        A.run(1, G); B.run(1, G)
        torch.cuda.synchronize()
        for nm, h, r in (("A", A, ra), ("B", B, rb)):
            s = h.state()
            bad = [n for n in NAMES if not torch.equal(s[n], r[k][n])]
            if bad:
                nbad += 1
                print("step", k, nm, "WRONG:", "; ".join("%s: %s" % (n, where(s[n], r[k][n])) for n in bad), flush=True)
    print(prec, precB, "graph" if G else "eager", "wrong (handle, step) pairs:", nbad, "of", 2 * NSTEP, flush=True)
elif os.environ.get("MODE") == "skip":
    # which class of B's kernels does it take?  B's step graph is captured with some helper classes of mmdm.hip skipped (debug build:
    # dbg_skip bits 1 fp32 GEMM, 2 fp32 attention, 4 bf16 GEMM, 8 fp8 GEMM, 16 split GEMM, 32 plane attention, 64 bf16 attention, 128 AdaLN);
    # B's numbers are garbage then, A's are the ones compared
    from mixermdm_amd._lib import diag
    def seqA():
        A.begin(*ia); out = []
        s0 = A.state(); out.append({k: s0[k].clone() for k in ("x", "x2")})
        for k in range(NSTEP):
            A.run(1, True); s = A.state()
            out.append({n: s[n].clone() for n in NAMES})
        return out
    ra = seqA()
    ALL = 255
    masks = [("everything runs", 0), ("nothing but the geometry / copy kernels", ALL)] + [("only class %d" % b, ALL & ~b) for b in (1, 2, 4, 8, 16, 32, 64, 128)] + \
            [("all but class %d" % b, b) for b in (1, 4, 16, 32, 64, 128)]
    for j, (what, m) in enumerate(masks):
        ibm = tuple(t.cuda() for t in synthetic_inputs(1, 263 - j, seed_cond=263, seed_x=264))
        diag("dbg_skip", 0)
        B.begin(*ibm); B.synchronize()
        diag("dbg_skip", m); B.run(1, True); diag("dbg_skip", 0)
        A.begin(*ia); torch.cuda.synchronize()
        nbad = 0
        for k in range(1, NSTEP + 1):
            s = A.state(); s["x"].copy_(ra[k - 1]["x"]); s["x2"].copy_(ra[k - 1]["x2"])
            torch.cuda.synchronize()
            A.run(1, True); B.run(1, True)
            torch.cuda.synchronize()
            s = A.state()
            nbad += int(any(not torch.equal(s[n], ra[k][n]) for n in NAMES))
        print("A %s beside B %s, B's graph with %s: A wrong in %d of %d steps" % (prec, precB, what, nbad, NSTEP), flush=True)
elif os.environ.get("MODE") == "gemm_aggr":
    # the aggressor reduced to ONE stand-alone GEMM launched over and over on a side stream beside A's step
    from mixermdm_amd import ops
    def seqA():
        A.begin(*ia); out = []
        s0 = A.state(); out.append({k: s0[k].clone() for k in ("x", "x2")})
        for k in range(NSTEP):
            A.run(1, True); s = A.state()
            out.append({n: s[n].clone() for n in NAMES})
        return out
    ra = seqA()
    side = torch.cuda.Stream()
    g = torch.Generator().manual_seed(3)
    def mk(kind, M, N, K, **kw):
        x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) * 0.03).cuda(); b = torch.randn(N, generator=g).cuda()
        if kind == "split":
            xs, ws = ops.split_f32(x), ops.split_f32(w)
            if kw.get("packed", True): ws = ops.split_pack_weight(ws)
            return lambda: ops.linear_split(xs, ws, b, split_out=kw.get("split_out", False), packed=kw.get("packed", True))
        if kind == "bf16":
            xb, wb = ops.to_bf16(x), ops.to_bf16(w)
            if kw.get("packed", True): wb = ops.pack_weight_frag(wb)
            return lambda: ops.linear_bf16(xb, wb, b, packed=kw.get("packed", True))
        return lambda: ops.linear(x, w, b)
    cfgs = [("fp32 GEMM 1052x1024x1024 (control)", ("f32", 1052, 1024, 1024), {}),
            ("split packed 1052x1024x1024", ("split", 1052, 1024, 1024), {}),
            ("split packed 1024x1024x1024 (whole tiles)", ("split", 1024, 1024, 1024), {}),
            ("split packed 1052x1024x1024, plane output", ("split", 1052, 1024, 1024), {"split_out": True}),
            ("split packed 1052x3072x1024", ("split", 1052, 3072, 1024), {}),
            ("split packed 4096x1024x1024 (128-row tiles)", ("split", 4096, 1024, 1024), {}),
            ("split UNPACKED 1052x1024x1024", ("split", 1052, 1024, 1024), {"packed": False}),
            ("bf16 packed 1052x1024x1024", ("bf16", 1052, 1024, 1024), {}),
            ("bf16 UNPACKED 1052x1024x1024", ("bf16", 1052, 1024, 1024), {"packed": False})]
    NAG = int(os.environ.get("NAGGR", "300"))
    for what, (kind, M, N, K), kw in cfgs:
        f = mk(kind, M, N, K, **kw)
        with torch.cuda.stream(side):
            want = f().clone()
        A.begin(*ia); torch.cuda.synchronize()
        nbad = 0; gbad = 0; over = 0
        for k in range(1, NSTEP + 1):
            s = A.state(); s["x"].copy_(ra[k - 1]["x"]); s["x2"].copy_(ra[k - 1]["x2"])
            torch.cuda.synchronize()
            A.run(1, True)
            with torch.cuda.stream(side):
                for _ in range(NAG): o = f()
            over += int(not A.stream.query())
            torch.cuda.synchronize()
            s = A.state()
            nbad += int(any(not torch.equal(s[n], ra[k][n]) for n in NAMES))
            gbad += int(not torch.equal(o, want))
        print("A %s beside %d x %s: A wrong in %d of %d steps (A still running when the last GEMM was queued: %d); the GEMM's own last result wrong: %d" % (prec, NAG, what, nbad, NSTEP, over, gbad), flush=True)
elif os.environ.get("MODE") == "canary":
    # tools/canary.hip beside B's step: known values held in LDS / registers / global memory / an LDS-DMA image, re-checked for a few ms
    import ctypes as C
    _cl = os.environ.get("CANARY_LIB", "libcanary.so")          # a file name under build/ (not shipped to the GPU box by gpurun) or a path (e.g. variants/libcanary.so)
    lib = C.CDLL(_cl if os.sep in _cl else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", _cl))
    side = torch.cuda.Stream()
    nwg = int(os.environ.get("CANARY_WG", "512")); lds = int(os.environ.get("CANARY_LDS", "32768")); spin = int(os.environ.get("CANARY_US", "5000"))
    report = torch.zeros(80, dtype=torch.int32, device="cuda"); gbuf = torch.zeros(nwg * 4096, dtype=torch.int32, device="cuda")
    pattern = torch.randint(-2**31, 2**31 - 1, (32768,), dtype=torch.int32, device="cuda")
    tin = torch.randn(nwg * 256, 6, device="cuda"); rec = torch.zeros(256 * 8, device="cuda")
    B.begin(*ib); A.begin(*ia); torch.cuda.synchronize()
    aggr = {}
    if os.environ.get("CANARY_AGGR") == "1":
        from mixermdm_amd import ops
        g = torch.Generator().manual_seed(3)
        M_, N_, K_ = 8192, 1024, 1024
        x = torch.randn(M_, K_, generator=g).cuda(); w = (torch.randn(N_, K_, generator=g) * 0.03).cuda(); bb = torch.randn(N_, generator=g).cuda()
        xs, ws = ops.split_f32(x), ops.split_f32(w); wsp = ops.split_pack_weight(ws)
        xb, wb = ops.to_bf16(x), ops.to_bf16(w); wbp = ops.pack_weight_frag(wb)
        t32a = torch.randn(4096, 4096, device="cuda"); t32b = torch.randn(4096, 4096, device="cuda")
        tba, tbb = t32a.bfloat16(), t32b.bfloat16(); tha, thb = t32a.half(), t32b.half()
        aux = torch.cuda.Stream()
        if os.environ.get("CU_SPLIT") == "1":
            # victim and aggressor on DISJOINT halves of the CUs (hipExtStreamCreateWithCUMask): does the effect need a shared CU?
            hip = C.CDLL("libamdhip64.so")
            def masked(words):
                st_ = C.c_void_p(); arr = (C.c_uint32 * 8)(*words)
                rc_ = hip.hipExtStreamCreateWithCUMask(C.byref(st_), 8, arr); assert rc_ == 0, rc_
                return torch.cuda.ExternalStream(st_.value)
            lo, hi = 0x0000FFFF, 0xFFFF0000
            side = masked([lo] * 8); aux = masked([hi] * 8)
            print("victim on CU mask %08x x 8, aggressor on %08x x 8" % (lo, hi), flush=True)
        def loop(f, n):
            def go():
                with torch.cuda.stream(aux):
                    for _ in range(n): f()
            return go
        aggr = {"library fp32 GEMM 8192x1024x1024": loop(lambda: ops.linear(x, w, bb), 150),
                "library split GEMM (3 x fp16 MFMA), packed W": loop(lambda: ops.linear_split(xs, wsp, bb, packed=True), 300),
                "library split GEMM, W through LDS": loop(lambda: ops.linear_split(xs, ws, bb), 300),
                "library bf16 GEMM, packed W": loop(lambda: ops.linear_bf16(xb, wbp, bb, packed=True), 400),
                "torch fp32 matmul 4096^3": loop(lambda: t32a @ t32b, 8),
                "torch bf16 matmul 4096^3": loop(lambda: tba @ tbb, 60),
                "torch fp16 matmul 4096^3": loop(lambda: tha @ thb, 60),
                "torch fp32 elementwise (64M)": loop(lambda: t32a * 1.5 + t32b, 300)}
        if os.environ.get("CANARY_VARIANTS") == "1":
            from mixermdm_amd._lib import diag
            tlbuf = torch.zeros(64 * 8 * 4 * 8 * 2, dtype=torch.int64, device="cuda")
            def with_diag(pairs, f, n):
                def go():
                    for k_, v_ in pairs: diag(k_, v_[0])
                    with torch.cuda.stream(aux):
                        for _ in range(n): f()
                    for k_, v_ in pairs: diag(k_, v_[1])
                return go
            fs = lambda: ops.linear_split(xs, wsp, bb, packed=True)
            fso = lambda: ops.linear_split(xs, wsp, bb, packed=True, split_out=True)
            fb = lambda: ops.linear_bf16(xb, wbp, bb, packed=True)
            aggr = {"split packed, default": with_diag([], fs, 300),
                    "split packed, direct epilogue (split_tst 0)": with_diag([("split_tst", (0, 1))], fs, 300),
                    "split packed, timeline instantiation (stamps + sched_barrier in the loop)": with_diag([("split_timeline", (tlbuf.data_ptr(), 0))], fs, 300),
                    "split packed, plane output": with_diag([], fso, 300),
                    "bf16 packed, default": with_diag([], fb, 400),
                    "bf16 packed, direct epilogue (bf16_tst 0)": with_diag([("bf16_tst", (0, 1))], fb, 400)}
        if os.environ.get("CANARY_MICRO") == "1":
            sink = torch.zeros(16, device="cuda")
            def micro(kind, iters):
                def go():
                    rc_ = lib.aggressor_launch(kind, C.c_void_p(sink.data_ptr()), C.c_void_p(pattern.data_ptr()), 1024, iters, C.c_void_p(aux.cuda_stream)); assert rc_ == 0, rc_
                return go
            aggr = {"micro: fp16 MFMAs only": micro(1, 12000), "micro: buffer loads into registers only": micro(2, 12000), "micro: fp16 MFMAs fed by buffer loads": micro(3, 12000),
                    "micro: LDS-DMA only": micro(4, 12000), "micro: fp16 MFMAs + LDS-DMA": micro(5, 12000), "micro: buffer loads + LDS-DMA": micro(6, 12000),
                    "micro: fp16 MFMAs + buffer loads + LDS-DMA": micro(7, 12000), "micro: fp32 MFMAs only": micro(8, 3000), "micro: fp32 MFMAs fed by buffer loads": micro(10, 3000),
                    "micro: fp32 MFMAs + LDS-DMA": micro(12, 3000)}
        for f in aggr.values(): f()
        torch.cuda.synchronize()
    # CANARY_HSACO=<code object>: the canary kernels from a hand-assembled build (tools/canary.hip compiled to assembly, e.g. with an s_nop behind every
    # packed-fp32 instruction, assembled with clang -x assembler and linked with ld.lld) instead of build/libcanary.so
    mfn = None
    if os.environ.get("CANARY_HSACO"):
        hipm = C.CDLL("libamdhip64.so")
        mod = C.c_void_p(); rc_ = hipm.hipModuleLoad(C.byref(mod), os.environ["CANARY_HSACO"].encode()); assert rc_ == 0, rc_
        mfn = C.c_void_p()
        rc_ = hipm.hipModuleGetFunction(C.byref(mfn), mod, b"_Z19canary_chain_kernelPjPKfi" if os.environ.get("CANARY_CHAIN") == "1" else b"_Z19canary_trans_kernelPjPKfi"); assert rc_ == 0, rc_
    only = os.environ.get("CANARY_ONLY")
    if only:
        aggr = {k: v for k, v in aggr.items() if only in k}
    for what in (["alone"] if only else ["alone", "beside B's step", "beside A's and B's steps"]) + list(aggr) + ["alone"]:
        tot = torch.zeros(80, dtype=torch.int64)
        for k in range(NSTEP):
            report.zero_(); torch.cuda.synchronize()
            if os.environ.get("CANARY_PK") == "1":
                rec.zero_()
                rc = lib.canary_pk_launch(C.c_void_p(report.data_ptr()), C.c_void_p(rec.data_ptr()), C.c_void_p(tin.data_ptr()), nwg, spin, C.c_void_p(side.cuda_stream))
            elif mfn is not None:
                a0, a1, a2 = C.c_void_p(report.data_ptr()), C.c_void_p(tin.data_ptr()), C.c_int(spin)
                params = (C.c_void_p * 3)(C.addressof(a0), C.addressof(a1), C.addressof(a2))
                rc = hipm.hipModuleLaunchKernel(mfn, nwg, 1, 1, 256, 1, 1, 0, C.c_void_p(side.cuda_stream), params, None)
            elif os.environ.get("CANARY_CHAIN") == "1":
                rc = lib.canary_chain_launch(C.c_void_p(report.data_ptr()), C.c_void_p(tin.data_ptr()), nwg, spin, C.c_void_p(side.cuda_stream))
            elif os.environ.get("CANARY_OPS") == "1":
                rc = lib.canary_ops_launch(C.c_void_p(report.data_ptr()), C.c_void_p(tin.data_ptr()), nwg, spin, C.c_void_p(side.cuda_stream))
            elif os.environ.get("CANARY_TRANS") == "1":
                rc = lib.canary_trans_launch(C.c_void_p(report.data_ptr()), C.c_void_p(tin.data_ptr()), nwg, spin, C.c_void_p(side.cuda_stream))
            else:
                rc = lib.canary_launch(C.c_void_p(report.data_ptr()), C.c_void_p(gbuf.data_ptr()), C.c_void_p(pattern.data_ptr()), nwg, lds, spin, C.c_void_p(side.cuda_stream))
            assert rc == 0, rc
            if what in aggr:
                aggr[what]()
            elif what != "alone":
                if "A's" in what: A.run(1, G)
                B.run(2, G)
            torch.cuda.synchronize()
            tot += report.cpu().to(torch.int64)
        if os.environ.get("CANARY_PK") == "1":
            print("pk canary %s: %d mismatches of the explicit v_pk_mul_f32 ... op_sel_hi:[1,0] in %d x 256 evaluations" % (what, tot[5], tot[6]), flush=True)
            r_ = rec.cpu().view(-1, 8)
            for row in r_[:min(int(report[5].item()), 10)]:
                a0, a1, b0, b1, d0, d1, lane, it = [float(v) for v in row]
                import struct
                f32 = lambda x: struct.unpack("f", struct.pack("f", x))[0]
                cands = {"a.lo*b.lo": f32(a0 * b0), "a.hi*b.lo": f32(a1 * b0), "a.lo*b.hi": f32(a0 * b1), "a.hi*b.hi": f32(a1 * b1), "sentinel lo": 777.0, "sentinel hi": 888.0, "zero": 0.0}
                name = lambda v: next((k for k, c in cands.items() if c == v), "?")
                print("    lanes wrong %d  a = (%.6g, %.6g) b = (%.6g, %.6g)  got (%.6g = %s, %.6g = %s)  expected (%.6g, %.6g)" % (int(lane), a0, a1, b0, b1, d0, name(d0), d1, name(d1), cands["a.lo*b.lo"], cands["a.hi*b.lo"]), flush=True)
            continue
        if os.environ.get("CANARY_CHAIN") == "1":
            nm = "b1x b1y b1z dt b2x b2y b2z b3x b3y b3z qw qx qy qz nrm half s1 ax ay az ang2 s2 r i j k two_s o0 o1 o2 o3 o4 o5".split()
            print("chain canary %s: %d x 256 evaluations; first intermediate that moved: %s" % (what, tot[6], ", ".join("%s %d" % (nm[i], tot[32 + i]) for i in range(33) if tot[32 + i]) or "none"), flush=True)
            continue
        if os.environ.get("CANARY_OPS") == "1":
            names = ("fma chain", "division", "sqrtf", "sinf", "cosf", "atan2f", "expf", "erff", "integer mix", "v_rcp_f32", "v_sin_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")
            print("ops canary %s: %d x 256 evaluations; moved: %s" % (what, tot[6], ", ".join("%s %d" % (nm, tot[8 + i]) for i, nm in enumerate(names))), flush=True)
            continue
        print("canary (%d workgroups, %d KB LDS, %d us) %s [%s]: mismatches LDS %d, registers %d, global %d, LDS-DMA image %d; %d checks; rotation round trips that moved: %d of %d x 256 (input registers changed %d, reference registers changed %d, neither %d)" %
              (nwg, lds // 1024, spin, what, precB, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], tot[6], tot[7], tot[20], tot[21]), flush=True)
elif os.environ.get("MODE") == "which":
    # which of A's buffers differ after an overlapped step?  (debug build: snap_handle / diff_handle list A's allocations by name)
    from mixermdm_amd._lib import diag
    S_ = A.schedule.num_timesteps
    A.begin(*ia); B.begin(*ib)
    x_prev = {k: v.clone() for k, v in A.state().items() if k in ("x", "x2")}
    for k in range(1, NSTEP + 1):
        def put():
            s = A.state(); s["x"].copy_(x_prev["x"]); s["x2"].copy_(x_prev["x2"]); torch.cuda.synchronize()
            A.seek(S_ - k); A.synchronize()
        put(); A.run(1, G); A.synchronize()
        good = {n: v.clone() for n, v in A.state().items()}
        diag("snap_handle", A.h.value)
        if os.environ.get("POISON") == "1":
            put(); diag("poison_scratch", A.h.value); A.run(1, G); A.synchronize()
            st_ = A.state()
            print("step", k, "A alone after poisoning its scratch:", "eq" if all(torch.equal(st_[n], good[n]) for n in NAMES) else "DIFF (nan: %s)" % bool(torch.isnan(st_["model_out"]).any()), flush=True)
            put(); diag("poison_scratch", A.h.value)
        else:
            put()
        A.run(1, G); B.run(1, G); torch.cuda.synchronize()
        st_ = A.state()
        if os.environ.get("POISON") == "1":
            print("step", k, "beside B after poisoning: NaN in model_out %s, x %s" % (bool(torch.isnan(st_["model_out"]).any()), bool(torch.isnan(st_["x"]).any())), flush=True)
        if any(not torch.equal(st_[n], good[n]) for n in NAMES):
            print("step", k, "A wrong beside B; A's allocations that differ from the same step run alone:", flush=True)
            diag("diff_handle", A.h.value)
        else:
            print("step", k, "A right", flush=True)
        x_prev = {"x": good["x"], "x2": good["x2"]}
elif os.environ.get("MODE") == "victim_torch":
    # the other way round: a torch matmul (rocBLAS: an LDS user) as the victim beside the handle's steps
    side = torch.cuda.Stream()
    m1 = torch.randn(2048, 2048, device="cuda"); m2 = torch.randn(2048, 2048, device="cuda")
    want = m1 @ m2
    torch.cuda.synchronize()
    for rnd in range(3):
        out = torch.empty_like(ia[1])
        A.enqueue(ia[0], ia[1], out, use_graph=G)
        res = []
        with torch.cuda.stream(side):
            for _ in range(300):
                res.append(m1 @ m2)
        busy = not A.stream.query()
        torch.cuda.synchronize()
        print(prec, "torch matmul beside the handle (handle still running at the end: %s): %d of %d products wrong; handle:" % (busy, sum(int(not torch.equal(r, want)) for r in res), len(res)),
              "eq" if torch.equal(out, refA_full) else "DIFF", flush=True)
else:
    kind = os.environ.get("AGGR", "matmul")
    side = torch.cuda.Stream()
    if kind == "matmul":
        m1 = torch.randn(4096, 4096, device="cuda"); m2 = torch.randn(4096, 4096, device="cuda")
    else:
        m1 = torch.randn(64 << 20, device="cuda"); m2 = torch.randn(64 << 20, device="cuda")
    torch.cuda.synchronize()
    for rnd in range(4):
        out = torch.empty_like(ia[1])
        with torch.cuda.stream(side):
            for _ in range(int(os.environ.get("NAGGR", "400"))):
                m3 = (m1 @ m2) if kind == "matmul" else (m1 + m2)
        A.enqueue(ia[0], ia[1], out, use_graph=G)
        A.synchronize()
        busy = not side.query()
        torch.cuda.synchronize()
        print(prec, "victim beside", kind, "(aggressor still running at the end: %s):" % busy, "eq" if torch.equal(out, refA_full) else "DIFF " + where(out, refA_full), flush=True)
