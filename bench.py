#!/usr/bin/env python3
"""bench.py -- generated motions/sec of the 2-person MixerMDM denoising loop (1000-step DDIM eta=0, T=300) on N MI355X.

A "step" is one DDIM step of the whole hot path (two denoisers + geometry + Influence mixer + blend + two-chain
update) over one batch of B motions per GPU (BASELINE.json configs[2]: B=16, T=300, fp32).  value = N*B motions per
1000 steps of the measured per-step time.  One JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
    python bench.py --gpus 8            (no launcher: starts the N ranks itself as fresh child processes, before any GPU call)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X dense bf16 MFMA peak (same guide)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X fp32 matrix peak (/opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters)


def algorithmic_flops_per_motion_step(T, D=1024, F=2048, L=8, Dm=512, Fm=1024, Lm=4, single=False):
    """478.4 GFLOP at T=300 (2-person); single-person 2*d1 + AdaLN = 55.7 GFLOP at T=196."""
    d1 = 2 * T * (L * (4 * D * D + 2 * D * F + 2 * (T + 1) * D) + 2 * 262 * D)
    if single:
        return 2 * d1 + 2 * L * 2 * 2 * D * D * 2
    d2 = 4 * T * (L * (8 * D * D + 2 * D * F + 4 * (T + 1) * D) + 2 * 262 * D)
    inf = 2 * T * (Lm * (8 * Dm * Dm + 2 * Dm * Fm + 4 * (T + 1) * Dm) + 23 * Dm)
    ada = 2 * L * 2 * 2 * D * D * 4 + 2 * L * 4 * 2 * 2 * D * D * 2 + 2 * Lm * 4 * 2 * Dm * Dm * 4
    return 4 * d1 + 2 * d2 + 4 * inf + 8 * (2 * T * 262 * Dm) + ada


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["mixer", "single"], default="mixer",
                    help="mixer = BASELINE configs[2] (2-person MixerMDM, the headline metric); single = configs[1] (single-person in2IN, T=196, B=32)")
    ap.add_argument("--batch", type=int, default=None, help="motions per GPU (weak scaling); default 16 (mixer) / 32 (single)")
    ap.add_argument("--frames", type=int, default=None, help="default 300 (mixer) / 196 (single)")
    ap.add_argument("--precision", choices=["fp32", "fp32_split", "bf16", "bf16_fp8"], default="fp32",
                    help="fp32 = the parity path and the headline metric; bf16 / bf16_fp8 = BASELINE configs[4] path (bf16 GEMM operands, fp32 accumulate; "
                         "bf16_fp8: QKV / cross-attention input / FFN GEMMs on fp8 e4m3 operands)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-alt", action="store_true", help="skip the extra fp32_split measurement reported next to the fp32 headline (N=1 only)")
    ap.add_argument("--no-full-loop", action="store_true", help="skip the real 1000-step sample() from x_T to x_0 reported as `full_loop`")
    ap.add_argument("--no-clock", action="store_true", help="skip the in-loop shader-clock measurement (301 extra GEMM launches; tools/profile_round.sh passes it so that "
                                                             "the committed kernel traces hold the step's launches only)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous check without a GPU: ranks meet on gloo, time a barrier, rank 0 prints a line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Bare `python bench.py --gpus N`: start the N ranks as FRESH child processes (one per GPU) through torch.distributed.run and
        # relay their output.  This parent has made no GPU call (torch is not even imported yet) and never replaces itself: it waits
        # for the launcher and exits with its status.
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); run `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if args.dry_run:
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([float(rank)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_rank_seen": int(t.item())}), flush=True)
        return
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # MMDM_BENCH_FORCE_DIST=1 under torchrun --nproc-per-node 1 exercises the RCCL code path (init, broadcast, barrier, all_reduce) on one GPU
    force_dist = world == 1 and os.environ.get("MMDM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    from mixermdm_amd.sampler import Sampler
    from mixermdm_amd.synthetic import synthetic_state_dict, mixer_shapes, synthetic_stats, synthetic_inputs, FULL_DIMS
    from mixermdm_amd.distributed import broadcast_state_dict

    single = args.workload == "single"
    # motions per GPU: configs[2] = 16 on one GPU (the headline); configs[3] = 256 over 8 GPUs = 32 per GPU; configs[1] (single) = 32
    B = args.batch or (32 if single else (32 if world == 8 else 16))
    T = args.frames or (196 if single else 300)
    S = 1000
    # weights: rank 0 draws them on the host, ONE RCCL broadcast of the packed 1.46 GB vector over xGMI (no other collective on the path)
    shapes = mixer_shapes(single_only=single, **FULL_DIMS)
    sd_cpu = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, single_only=single, **FULL_DIMS) if rank == 0 else None
    sd = broadcast_state_dict(sd_cpu, shapes, src=0, device=device)
    stats = synthetic_stats()
    smp = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, single_only=single, precision=args.precision, **FULL_DIMS)
    smp.load_state_dict(sd)
    if not single:
        smp.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    smp.prepare()
    smp.set_schedule("ddim1000")
    del sd
    cond, xT = synthetic_inputs(B, T, seed_cond=1 + 1000 * rank, seed_x=2 + 1000 * rank, single=single)   # each rank = its own shard of the batch
    cond, xT = cond.to(device), xT.to(device)
    use_graph = not args.no_graph
    if args.warmup + args.steps > S:
        raise SystemExit("warmup + steps must be <= 1000")

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    smp.begin(cond, xT)                      # inputs resident in HBM before the timed region
    smp.run(args.warmup, use_graph)          # untimed warm-up (includes the graph capture)
    barrier()
    t0 = time.perf_counter()
    smp.run(args.steps, use_graph)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    per_rank = [dt]
    if dist.is_initialized():
        every = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(every, tmax)
        per_rank = [float(v.item()) for v in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    rank_ms = {"min": round(min(per_rank) / args.steps * 1e3, 3), "max": round(max(per_rank) / args.steps * 1e3, 3)}
    finite = bool(torch.isfinite(smp.state()["x"]).all().item())

    # dominant kernel: the fp32 MFMA GEMM -- live HIP-event timing of every launch, eager, on the handle's stream
    roof = None
    if rank == 0 and args.profile_steps > 0 and args.warmup + args.steps + args.profile_steps <= S:
        smp.profile(True)
        smp.run(args.profile_steps, use_graph=False)
        g_ms, g_n, g_fl, g_by = smp.profile_read(0)
        a_ms, a_n, a_fl, _ = smp.profile_read(1)
        smp.profile(False)
        ach = g_fl / (g_ms * 1e-3) / 1e12
        # fp32_split executes SIX bf16 MFMAs per algorithmic multiply-add block: its roof is the dense bf16 peak / 6
        # bf16_fp8: the non-scaled fp8 MFMA (v_mfma_f32_32x32x16_fp8_fp8) issues at the bf16 rate on gfx950 (MI355X_MICROARCH.md, Matrix
        # cores), and a third of the mode's GEMM launches are bf16: both priced against the dense bf16 peak
        peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "fp32_split": round(PEAK_BF16_MFMA_TFLOPS / 6, 1), "bf16_fp8": PEAK_BF16_MFMA_TFLOPS}[args.precision]
        kname = {"fp32": "gemm_glds_kernel<..., PIPE_=1> (v_mfma_f32_32x32x2_f32; software-pipelined LDS-DMA ring: 128x128 tiles x 5 stages, 128x64 x 4 stages when N = 2048 or N, K <= 512; all instantiations of a step averaged)",
                 "bf16": "gemm_bf16w_kernel (v_mfma_f32_32x32x16_bf16; weights in MFMA fragment order fetched straight from global memory, A through three LDS-DMA stages, 128x256 tiles, K step 128 bytes)",
                 "fp32_split": "gemm_splitw_kernel (fp32 result from 6 x v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 operand splits; weights in MFMA fragment order fetched straight from global memory, 128x128 tiles, two workgroups per CU; peak = 2500/6 algorithmic TFLOP/s)",
                 "bf16_fp8": "gemm_bf16_kernel<ET=1> (v_mfma_f32_32x32x16_fp8_fp8: e4m3 operands, per-row / per-output-channel scales, fp32 accumulate) "
                             "+ gemm_bf16w_kernel (bf16, packed weights) for the attention output projections"}[args.precision]
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": measured_traffic(single, args.precision),
                "algorithmic_mb_per_launch": round(g_by / g_n / 1e6, 1),
                "launches_per_step": g_n // args.profile_steps, "avg_launch_us": round(g_ms * 1e3 / g_n, 2),
                "gflop_per_launch": round(g_fl / g_n / 1e9, 3), "gemm_ms_per_step": round(g_ms / args.profile_steps, 3),
                "attention": {"achieved": round(a_fl / (a_ms * 1e-3) / 1e12, 2), "ms_per_step": round(a_ms / args.profile_steps, 3),
                              "launches_per_step": a_n // args.profile_steps}}
        try:
            roof["clock"] = None if args.no_clock else loop_clock(args.precision, 4 * B * T if not single else 2 * B * T, peak, ach)
        except Exception as e:          # a diagnostic next to the measurement, never a reason to lose the line
            roof["clock"] = {"error": repr(e)}
    if dist.is_initialized():
        dist.barrier()

    # Same workload, same inputs, in the fp32-split mode (fp32-accurate GEMMs on the bf16 matrix cores: DESIGN.md 6c).  Reported beside the
    # headline, never as it: the headline stays the native-fp32 parity path that the reference-captured goldens pin end to end.
    alt = None
    if world == 1 and args.precision == "fp32" and not args.no_alt and not single:
        alt_smp = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, single_only=single, precision="fp32_split", **FULL_DIMS)
        alt_smp.load_state_dict(sd_cpu)
        alt_smp.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
        alt_smp.prepare()
        alt_smp.set_schedule("ddim1000")
        alt_smp.begin(cond, xT)
        alt_smp.run(args.warmup, use_graph)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        alt_smp.run(args.steps, use_graph)
        torch.cuda.synchronize()
        a_ms = (time.perf_counter() - t1) / args.steps * 1e3
        # how far the two modes are apart after the same number of steps from the same state (fp32 rounding noise level)
        smp.begin(cond, xT); smp.run(2, use_graph)
        alt_smp.begin(cond, xT); alt_smp.run(2, use_graph)
        xa, xb = smp.state()["x2"], alt_smp.state()["x2"]
        rel = float(((xa - xb).pow(2).mean().sqrt() / xa.pow(2).mean().sqrt()).item())
        alt = {"mode": "fp32_split (exact 3-way bf16 operand split, six bf16 MFMAs per product, fp32 accumulate)", "ms_per_step": round(a_ms, 3),
               "value": round(B / (a_ms * 1e-3 * S), 5), "unit": "motions/s", "rel_rms_vs_fp32_after_2_steps": rel,
               "achieved_tflops_algorithmic": round(algorithmic_flops_per_motion_step(T, single=single) * B / (a_ms * 1e-3) / 1e12, 2)}
        # its own roofline: live HIP-event pairs around every GEMM launch of an eager pass, against 2500 / 6 algorithmic TFLOP/s
        if args.profile_steps > 0:
            alt_smp.profile(True)
            alt_smp.run(args.profile_steps, use_graph=False)
            g_ms, g_n, g_fl, g_by = alt_smp.profile_read(0)
            a2_ms, a2_n, a2_fl, _ = alt_smp.profile_read(1)
            alt_smp.profile(False)
            pk = round(PEAK_BF16_MFMA_TFLOPS / 6, 1)
            ach2 = g_fl / (g_ms * 1e-3) / 1e12
            alt["roofline"] = {"bound": "mfma", "kernel": "gemm_splitw_kernel (fp32 result from 6 x v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 operand splits; packed weights straight from global memory, 128x128 tiles, two workgroups per CU)",
                               "achieved": round(ach2, 2), "peak": pk, "unit": "TFLOP/s", "frac": round(ach2 / pk, 4), "traffic": measured_traffic(single, "fp32_split"),
                               "algorithmic_mb_per_launch": round(g_by / g_n / 1e6, 1), "launches_per_step": g_n // args.profile_steps, "avg_launch_us": round(g_ms * 1e3 / g_n, 2),
                               "gemm_ms_per_step": round(g_ms / args.profile_steps, 3),
                               "attention": {"achieved": round(a2_fl / (a2_ms * 1e-3) / 1e12, 2), "ms_per_step": round(a2_ms / args.profile_steps, 3), "launches_per_step": a2_n // args.profile_steps}}
            try:
                alt["roofline"]["clock"] = None if args.no_clock else loop_clock("fp32_split", 4 * B * T, pk, ach2)
            except Exception as e:
                alt["roofline"]["clock"] = {"error": repr(e)}
        alt_smp.close()

    # The metric itself, not an extrapolation: one whole sample() from x_T to x_0 (S graph replays + the begin() set-up), every rank
    # on its own shard, wall time = max over ranks.
    full = None
    if not args.no_full_loop:
        barrier()
        t1 = time.perf_counter()
        out = smp.sample(cond, xT, use_graph=use_graph)
        barrier()
        fl = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
        if dist.is_initialized():
            dist.all_reduce(fl, op=dist.ReduceOp.MAX)
        wall = float(fl.item())
        full = {"steps": S, "wall_s": round(wall, 3), "motions_per_s": round(world * B / wall, 5), "outputs_finite": bool(torch.isfinite(out).all().item())}
        del out

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:       # reported at N=1 only (the other ranks of an N>1 run would idle behind it)
        cpu = cpu_baseline(sd_cpu, stats, T, args.cpu_steps, single)

    if rank == 0:
        flops = algorithmic_flops_per_motion_step(T, single=single)
        value = world * B / (ms_per_step * 1e-3 * S)
        wl = ("BASELINE configs[1]: single-person in2IN (individual denoiser, CFG 3.5), T=%d, ddim1000 (eta=0), batch %d per GPU, %s, random-init weights" % (T, B, args.precision)) if single else \
             ("BASELINE %s: 2-person MixerMDM (in2IN individual + in2IN interaction + Mixer mode 4, align, CFG 3.5), "
              "T=%d, ddim1000 (eta=0), batch %d per GPU, %s, random-init weights" % ("configs[3] (batch 256 over 8 GPUs = 32 per GPU)" if world == 8 and B == 32 else "configs[2]", T, B, {"fp32": "fp32", "fp32_split": "fp32 via exact 3-way bf16 operand split (six bf16 MFMAs per product)", "bf16": "bf16 GEMM operands / fp32 accumulate (configs[4]-style)", "bf16_fp8": "configs[4]: bf16 path with fp8 e4m3 QKV / FFN GEMM operands, fp32 accumulate"}[args.precision]))
        line = {
            "metric": "generated motions/sec (1000-step DDPM schedule sampled with DDIM eta=0, T=%d, %s)" % (T, "single-person" if single else "2-person"),
            "value": round(value, 5), "unit": "motions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_ranks": rank_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "fp32_split": "f32 (3xbf16 exact operand split, fp32 accumulate)", "bf16": "bf16", "bf16_fp8": "bf16 + fp8 e4m3 QKV/FFN operands"}[args.precision], "data": "synthetic",
            "config": {"workload": wl,
                       "batch_per_gpu": B, "frames": T, "sampler_steps": S, "hipgraph": use_graph, "parallelism": "batch-sharded x%d, no in-loop collective" % world},
            "achieved_tflops_algorithmic": round(flops * B * world / (ms_per_step * 1e-3) / 1e12, 2),
            "frac_of_f32_mfma_peak": round(flops * B / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if args.precision in ("fp32", "fp32_split") else None,
            "outputs_finite": finite,
            "full_loop": full, "roofline": roof, "cpu_baseline": cpu, "fp32_split": alt,
        }
        print(json.dumps(line), flush=True)
    smp.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def loop_clock(precision, M, peak, achieved):
    """Shader clock INSIDE the K loop of the dominant GEMM (diagnostic stamps: s_memtime beside the 100 MHz s_memrealtime) on one QKV-shaped
    launch (M x 3072 x 1024) after 300 warm launches (the clock needs ~30 ms of load to leave its idle ramp, then settles at what the power budget allows).  The guide's peaks are 2.4 GHz figures; a kernel that keeps the matrix pipes busy runs
    against the power-managed clock, so the same achieved rate is also quoted against the peak at the measured clock."""
    import ctypes as C
    import re
    import torch
    from mixermdm_amd import ops, load_library
    if precision not in ("fp32", "fp32_split"):
        return None
    lib = load_library()
    d = torch.device("cuda:0")
    N, K = 3072, 1024
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5; b = torch.randn(N, device=d)
    if precision == "fp32":
        call = lambda: ops.linear(x, w, b)
        lib.mmdmx_set_gemm_tail(0)          # one launch per call: the stamps are indexed by workgroup (the single-chain samplers split the last round off)
        for _ in range(300):
            call()
        # 10 words per workgroup (8 stamps + 2 cycle counters), sized for the smallest tile the dispatch can pick (64 x 64): the stamping
        # kernel writes without a bound check, so the buffer must cover every workgroup of this launch
        buf = torch.zeros(10 * ((M + 63) // 64) * ((N + 63) // 64), dtype=torch.int64, device=d)
        lib.mmdmx_set_gemm_stamps(C.c_void_p(buf.data_ptr()))
        call(); torch.cuda.synchronize()
        lib.mmdmx_set_gemm_stamps(C.c_void_p(0))
        lib.mmdmx_set_gemm_tail(-1)
        kern = lib.mmdm_last_gemm_kernel().decode()
        tm_, tn_ = [int(v) for v in re.search(r"<(\d+),(\d+)", kern).groups()]
        bm, bn = 32 * (tm_ // 10) * (tm_ % 10), 32 * (tn_ // 10) * (tn_ % 10)
        n = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        raw = buf.cpu()
        st, ck = raw[:8 * n].view(n, 8).double(), raw[8 * n:10 * n].view(n, 2).double()
        mhz = (100.0 * (ck[:, 1] - ck[:, 0]) / (st[:, 2] - st[:, 1]).clamp(min=1)).median().item()
    else:
        xs, ws = ops.split3(x), ops.split3(w)
        wp = ops.split_pack_weight(ws)
        call = lambda: ops.linear_split(xs, wp, b, packed=True)
        for _ in range(300):
            call()
        kern = lib.mmdm_last_gemm_kernel().decode()
        tm_, tn_ = [int(v) for v in re.search(r"<(\d+),(\d+)", kern).groups()]
        bm, bn, waves = 32 * (tm_ // 10) * (tm_ % 10), 32 * (tn_ // 10) * (tn_ % 10), (tm_ // 10) * (tn_ // 10)
        n = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        buf = torch.zeros(n * waves * 8, dtype=torch.int64, device=d)
        lib.mmdmx_set_split_timeline.argtypes = [C.c_void_p]
        lib.mmdmx_set_split_timeline(C.c_void_p(buf.data_ptr()))
        call(); torch.cuda.synchronize()
        lib.mmdmx_set_split_timeline(None)
        t = buf.view(n * waves, 8).double().cpu()
        mhz = (100.0 * t[:, 7] / t[:, 6].clamp(min=1)).median().item()
    pk = peak * mhz / 2400.0
    return {"shader_mhz_in_k_loop": round(mhz), "kernel": kern, "peak_at_that_clock": round(pk, 1), "frac_of_peak_at_that_clock": round(achieved / pk, 4),
            "how": "s_memtime / s_memrealtime between K-loop entry and exit of every workgroup (median) on one %dx%dx%d launch after 300 warm ones; "
                   "`peak` above is the guide's 2.4 GHz figure" % (M, N, K)}


def measured_traffic(single, precision="fp32"):
    """HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes of this workload and precision mode (profiles/gemm_traffic.json
    for fp32, profiles/gemm_traffic_<mode>.json otherwise; written by tools/pmc_summary.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams on gfx950).  bench.py cannot collect PMC counters
    itself.  The artefact records the hash of the kernel sources it was measured on: null when it is absent, for another workload, or
    was taken on different kernel sources than the ones this run executes."""
    from mixermdm_amd.build import sources_sha
    path = os.path.join(ROOT, "profiles", "gemm_traffic.json" if precision == "fp32" else "gemm_traffic_%s.json" % precision)
    if single or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("kernel_sources_sha") != sources_sha(precision):
        print("bench.py: %s was measured on other kernel sources (%s != %s): roofline.traffic = null" % (os.path.basename(path), rec.get("kernel_sources_sha"), sources_sha(precision)), file=sys.stderr)
        return None
    return rec["traffic_bytes_per_launch"]


def cpu_baseline(sd_cpu, stats, T, nsteps, single=False):
    """The oracle (a PyTorch-CPU port of the reference path, parity-pinned by tests/golden) timed on this host's cores:
    `nsteps` consecutive DDIM steps at B=1 after one untimed step, extrapolated to the 1000-step loop."""
    import torch
    from oracle import mixer as MX, schedule as OS
    from oracle.layers import pe_table
    from mixermdm_amd.synthetic import synthetic_inputs
    if single:
        return cpu_baseline_single(sd_cpu, T, nsteps)
    W = dict(sd_cpu)
    W["sequence_pos_encoder.pe"] = pe_table(512)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(1024)
    ostats = (stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    spec = MX.MixerSpec(d_heads=8, m_heads=8)
    cond, xT = synthetic_inputs(1, T)
    x, x2 = xT.clone(), xT.clone()
    with torch.no_grad():
        # thread-count calibration (one untimed step each): B=1 GEMMs are small, all cores is not always the fastest
        ncpu = os.cpu_count() or 1
        best, calib = (None, 1e30), {}
        # (all hardware threads is never the fastest for B=1 and can be pathological -- 265 s per step on a 256-thread host -- so the
        # calibration stops at 64; `host_threads` reports what the box has)
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, 999, x, x2, cond)
            dt1 = time.perf_counter() - t0
            calib[nt] = dt1
            if dt1 < best[1]:
                best = (nt, dt1)
        torch.set_num_threads(best[0])
        x, x2, _, _ = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, 999, x, x2, cond)
        t0 = time.perf_counter()
        for k in range(nsteps):
            x, x2, _, _ = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, 998 - k, x, x2, cond)
        dt = (time.perf_counter() - t0) / nsteps
        # SURVEY 8d also asks the GPU's own batch on the CPU: B = 16 is one more calibration (the larger GEMMs want more threads) and two
        # timed steps, extrapolated x1000 like the B = 1 figure; it says how much batch efficiency the CPU gets (the GPU needs the batch, the CPU barely gains)
        b16 = None
        if nsteps >= 4:
            c16, x16 = synthetic_inputs(16, T)
            runs = {}
            for nt in sorted({min(ncpu, c) for c in (32, 64)}):      # one step each (13 s on a 256-thread host): the faster one is the sample
                torch.set_num_threads(nt)
                t0 = time.perf_counter()
                MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, 999, x16, x16, c16)
                runs[nt] = time.perf_counter() - t0
            nt16 = min(runs, key=runs.get)
            b16 = {"s_per_step": round(runs[nt16], 3), "cores": nt16, "motions_per_s": round(16.0 / (runs[nt16] * 1000), 7),
                   "sample": "one DDIM step at B=16 (the GPU's batch) on %s threads each (%s), the faster one extrapolated x1000 steps"
                             % (" / ".join(str(k) for k in sorted(runs)), ", ".join("%d thr %.1f s" % kv for kv in sorted(runs.items())))}
    return {"value": round(1.0 / (dt * 1000), 7), "unit": "motions/s", "cores": best[0], "host_threads": ncpu, "kind": "port",
            "sample": "%d consecutive DDIM steps (i=998..) of the same workload at B=1, T=%d on the host CPU (PyTorch %s, fp32), "
                      "%.3f s/step, extrapolated x1000 steps; thread count calibrated over {8, 16, 32, 64} of the host's %d hardware threads: %d fastest (one step: %s)"
                      % (nsteps, T, torch.__version__, dt, ncpu, best[0], ", ".join("%d thr %.2f s" % kv for kv in sorted(calib.items()))),
            "s_per_step_b1": round(dt, 4), "b16": b16}


def cpu_baseline_single(sd_cpu, T, nsteps):
    import torch
    from oracle import mixer as MX, schedule as OS
    from oracle.layers import pe_table
    from mixermdm_amd.synthetic import synthetic_inputs
    W = dict(sd_cpu)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    sch = OS.make_schedule("cosine", 1000, "ddim1000")
    cond, x = synthetic_inputs(1, T, single=True)
    step = lambda i, x: MX.ddim_update(sch, i, x, MX.cfg_single(W, "denoiser1.", "individual", 3.5, x, torch.full((1,), sch.timestep_map[i], dtype=torch.long), cond, 8))
    with torch.no_grad():
        x = step(999, x)
        t0 = time.perf_counter()
        for k in range(nsteps):
            x = step(998 - k, x)
        dt = (time.perf_counter() - t0) / nsteps
    return {"value": round(1.0 / (dt * 1000), 7), "unit": "motions/s", "cores": torch.get_num_threads(), "host_threads": os.cpu_count(), "kind": "port",
            "sample": "%d consecutive DDIM steps of the single-person workload at B=1, T=%d on the host CPU (PyTorch %s, fp32), %.3f s/step, extrapolated x1000 steps" % (nsteps, T, torch.__version__, dt),
            "s_per_step_b1": round(dt, 4)}


if __name__ == "__main__":
    main()
