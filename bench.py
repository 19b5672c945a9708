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
PEAK_FP8_MFMA_TFLOPS = 5000.0     # MI355X dense fp8 peak: the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 the fp8 GEMMs issue (same guide)
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


def lib_stamp():
    """Which library this process ran, for every line: mmdm_version() of the loaded .so, its path (relative to the repo when inside it) and
    whether MMDM_LIB selected another build than the in-tree one (mixermdm_amd/_lib.py) -- a number measured on an A/B build must say so."""
    from mixermdm_amd._lib import load_library, lib_path, lib_override
    path = os.path.abspath(lib_path())
    return {"mmdm_version": load_library().mmdm_version().decode(), "lib": os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path,
            "lib_override": lib_override()}


def n1_same_batch(B, precision):
    """For the N > 1 lines (32 motions per GPU) the one-GPU figure AT THE SAME per-GPU batch: the default N = 1 line runs configs[2]'s 16 motions,
    whose motions/s/GPU differs from the 32-motion shard's by configuration, not by scaling.  Not measured in this run: the committed one-GPU
    bench line of the newest round that has one (profiles/rNN_bench_b32*.json), labelled as such."""
    import glob
    import re
    suffix = {"fp32": "", "fp32_split": "_split"}.get(precision)
    if suffix is None or B != 32:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_b32%s.json" % suffix)))
    files = [f for f in files if re.search(r"r\d\d_bench_b32%s\.json$" % suffix, f)]
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "per_gpu_batch": d.get("per_gpu_batch", 32), "n_gpus": 1,
                "source": os.path.relpath(files[-1], ROOT), "measured_in_this_run": False,
                "what": "one GPU at the SAME 32 motions per GPU (a committed one-GPU run of bench.py --batch 32, another box): the reference point for this line's scaling efficiency"}
    except Exception as e:
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["mixer", "single"], default="mixer",
                    help="mixer = BASELINE configs[2] (2-person MixerMDM, the headline metric); single = configs[1] (single-person in2IN, T=196, B=32)")
    ap.add_argument("--batch", type=int, default=None, help="motions per GPU (weak scaling); default 16 (mixer) / 32 (single)")
    ap.add_argument("--frames", type=int, default=None, help="default 300 (mixer) / 196 (single)")
    ap.add_argument("--sampler", default="ddim1000", help="sampling strategy (respacing of the 1000-step schedule): ddim1000 = the headline metric; "
                    "ddim50 = what the reference's own callers run (src/models/mixermdm.py:19, src/scripts/infer/mixermdm.py:73: B=1, T=299, ddim50)")
    ap.add_argument("--facade", action="store_true", help="build the sampler through the reference-API mirror (mixermdm_amd.models.MixerMDM, configs/models/*.yaml) and report "
                    "`facade`: one whole MixerMDM.forward(batch) / forward_test(batch) -- what src/scripts/infer/mixermdm.py and the evaluation datasets call")
    ap.add_argument("--no-whole-host", action="store_true", help="skip cpu_baseline.whole_host (k concurrent 16-thread B=1 oracle processes)")
    ap.add_argument("--cpu-worker", nargs=4, metavar=("THREADS", "STEPS", "FRAMES", "DIR"), default=None, help=argparse.SUPPRESS)
    ap.add_argument("--precision", choices=["fp32", "fp32_split", "bf16", "bf16_fp8"], default="fp32",
                    help="fp32 = the parity path and the headline metric; bf16 / bf16_fp8 = BASELINE configs[4] path (bf16 GEMM operands, fp32 accumulate; "
                         "bf16_fp8: QKV / cross-attention input / FFN GEMMs on fp8 e4m3 operands)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-alt", action="store_true", help="skip the extra fp32_split measurement reported next to the fp32 headline (N=1 only)")
    ap.add_argument("--no-full-loop", action="store_true", help="skip the real 1000-step sample() from x_T to x_0 reported as `full_loop`")
    ap.add_argument("--no-side", action="store_true", help="skip the two side measurements of the default line: `bf16_fp8` (configs[4]'s arithmetic on the same workload) and "
                                                            "`eval_items` (the reference's evaluation caller on 16 items: sequential loop vs ragged batches)")
    ap.add_argument("--side-eval-items", type=int, default=16, help="items of the default line's `eval_items` side measurement")
    ap.add_argument("--no-clock", action="store_true", help="skip the in-loop shader-clock measurement (301 extra GEMM launches; tools/profile_round.sh passes it so that "
                                                             "the committed kernel traces hold the step's launches only)")
    ap.add_argument("--eval-items", type=int, default=0, metavar="N",
                    help="the reference's evaluation caller instead of the headline step: N items with their own lengths (T uniform in [60, 300], ddim50 unless --sampler says "
                         "otherwise), one forward_test each (src/evaluation/datasets.py:100-116) -- timed as the reference's sequential loop, with 2 / 4 items in flight over "
                         "one weight set (mmdm_create_shared), and as ragged batches (mmdm_begin_ragged); reports items/s, roofline and cpu_baseline per strategy")
    ap.add_argument("--eval-max-rows", type=int, default=4800, help="--eval-items: frames per ragged batch (default 4800 = 16 x 300)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous check without a GPU: ranks meet on gloo, time a barrier, rank 0 prints a line")
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(*args.cpu_worker)

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

    if args.eval_items:
        return eval_items_bench(args, device, rank, world)

    single = args.workload == "single"
    # motions per GPU: configs[2] = 16 on one GPU (the headline); configs[3] = 256 over 8 GPUs = 32 per GPU, and the same 32 per GPU at
    # every N > 1 so that the points of a 2 / 4 / 8-GPU curve are one per-GPU workload (`per_gpu_batch` in the line); configs[1] (single) = 32
    B = args.batch or (32 if single else (32 if world > 1 else 16))
    T = args.frames or (196 if single else 300)
    # weights: rank 0 draws them on the host, ONE RCCL broadcast of the packed 1.46 GB vector over xGMI (no other collective on the path)
    shapes = mixer_shapes(single_only=single, **FULL_DIMS)
    sd_cpu = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, single_only=single, **FULL_DIMS) if rank == 0 else None
    sd = broadcast_state_dict(sd_cpu, shapes, src=0, device=device)
    stats = synthetic_stats()
    model = None
    if args.facade:
        # the reference's own entry point: MixerMDM(cfg, num_frames, sampling_strategy) over configs/models/MixerMDM.yaml; the facade owns the
        # Sampler (same handle type), so the timed region below is the same C call either way
        if single:
            raise SystemExit("--facade builds the two-person MixerMDM model (use --workload mixer)")
        from mixermdm_amd.configs import get_config
        from mixermdm_amd.models import MixerMDM
        model = MixerMDM(get_config(os.path.join(ROOT, "configs", "models", "MixerMDM.yaml")), num_frames=T, sampling_strategy=args.sampler, config_root=ROOT)
        model.precision = args.precision
        model.load_state_dict({"mixing." + k: v for k, v in sd.items()})
        model.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
        model = model.to(device).eval()
        smp = model._sampler_for(B, T)
    else:
        smp = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, single_only=single, precision=args.precision, **FULL_DIMS)
        smp.load_state_dict(sd)
        if not single:
            smp.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
        smp.prepare()
    S = smp.set_schedule(args.sampler).num_timesteps
    del sd
    cond, xT = synthetic_inputs(B, T, seed_cond=1 + 1000 * rank, seed_x=2 + 1000 * rank, single=single)   # each rank = its own shard of the batch
    cond, xT = cond.to(device), xT.to(device)
    use_graph = not args.no_graph
    if args.warmup + args.steps > S:
        raise SystemExit("warmup + steps must be <= %d (the steps of %s)" % (S, args.sampler))

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

    # dominant kernel of a precision mode: live HIP-event timing of every GEMM launch of an eager pass, on the handle's stream
    def live_roofline(smp, prec, with_clock=True):
        if args.warmup + args.steps + args.profile_steps > S:
            smp.begin(cond, xT)          # a short schedule (ddim50): profile the first steps of a fresh call
        smp.profile(True)
        smp.run(min(args.profile_steps, S), use_graph=False)
        g_ms, g_n, g_fl, g_by = smp.profile_read(0)
        a_ms, a_n, a_fl, _ = smp.profile_read(1)
        f_ms, f_n, f_fl, f_by = smp.profile_read(2)      # fp8-operand launches (bf16_fp8 only): their own class, their own peak
        p_ms, p_n, p_fl, _ = smp.profile_read(3)         # fp32 GEMMs of a low-precision handle (embeddings, conditioning, heads): priced against the fp32 roof
        smp.profile(False)
        # fp32_split executes THREE fp16 MFMAs per algorithmic multiply-add block: its roof is the dense 16-bit peak / 3.
        # bf16_fp8: two kinds of launches.  The QKV / cross-attention input / FFN GEMMs issue the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4
        # (e4m3 operands, unit E8M0 scales: twice the bf16 rate, 5 PFLOP/s dense) and are the dominant kernel: `achieved` / `peak` / `frac` are
        # theirs.  The attention output projections and embeddings are bf16 launches priced against 2.5 PFLOP/s in `bf16_launches`;
        # `frac_blended` = (time both classes would take at their own peaks) / (time they took).
        peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "fp32_split": round(PEAK_BF16_MFMA_TFLOPS / 3, 1), "bf16_fp8": PEAK_FP8_MFMA_TFLOPS}[prec]
        other = None
        fp32_side = None
        if p_n:
            p_ach = p_fl / (p_ms * 1e-3) / 1e12
            fp32_side = {"achieved": round(p_ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "frac": round(p_ach / PEAK_F32_MFMA_TFLOPS, 4), "launches_per_step": p_n // args.profile_steps,
                         "avg_launch_us": round(p_ms * 1e3 / p_n, 2), "ms_per_step": round(p_ms / args.profile_steps, 3),
                         "what": "the fp32-accurate side GEMMs of this mode: motion_embed on the fp32-split kernel (fp16 planes, three MFMAs per block), conditioning projections and heads on the fp32 MFMA kernel; priced against the fp32 MFMA peak"}
        if prec == "bf16_fp8" and f_n:
            b_ach = g_fl / (g_ms * 1e-3) / 1e12
            other = {"achieved": round(b_ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "frac": round(b_ach / PEAK_BF16_MFMA_TFLOPS, 4), "launches_per_step": g_n // args.profile_steps,
                     "avg_launch_us": round(g_ms * 1e3 / g_n, 2), "ms_per_step": round(g_ms / args.profile_steps, 3)}
            ideal_ms = (f_fl / PEAK_FP8_MFMA_TFLOPS + g_fl / PEAK_BF16_MFMA_TFLOPS) / 1e9
            blended = {"frac_blended": round(ideal_ms / (f_ms + g_ms), 4), "all_gemm_tflops": round((f_fl + g_fl) / ((f_ms + g_ms) * 1e-3) / 1e12, 2),
                       "all_gemm_frac_of_bf16_peak": round((f_fl + g_fl) / ((f_ms + g_ms) * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                       # every GEMM launch of the step incl. the fp32 embedding / conditioning / head GEMMs, as round 3 quoted it (0.31 at B = 16, 0.35 at B = 64)
                       "all_launches_tflops": round((f_fl + g_fl + p_fl) / ((f_ms + g_ms + p_ms) * 1e-3) / 1e12, 2),
                       "all_launches_frac_of_bf16_peak": round((f_fl + g_fl + p_fl) / ((f_ms + g_ms + p_ms) * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}
            g_ms, g_n, g_fl, g_by = f_ms, f_n, f_fl, f_by
        ach = g_fl / (g_ms * 1e-3) / 1e12
        kname = {"fp32": "gemm_glds_kernel<..., PIPE_=1> (v_mfma_f32_32x32x2_f32; software-pipelined LDS-DMA ring: 128x128 tiles x 5 stages, 128x64 x 4 stages when N = 2048 or N, K <= 512; all instantiations of a step averaged)",
                 "bf16": "gemm_bf16w_kernel (v_mfma_f32_32x32x16_bf16; weights in MFMA fragment order fetched straight from global memory, A through three LDS-DMA stages, 128x256 tiles, K step 128 bytes)",
                 "fp32_split": "gemm_splitw_kernel (fp32 result from 3 x v_mfma_f32_32x32x16_f16 on two-way fp16 operand splits, hi / lo accumulators; weights in MFMA fragment order fetched straight from global memory, 128x128 tiles, two workgroups per CU; peak = 2500/3 algorithmic TFLOP/s)",
                 "bf16_fp8": "gemm_bf16w_kernel<ET=fp8> / gemm_bf16_kernel<ET=fp8> (v_mfma_scale_f32_32x32x64_f8f6f4: e4m3 operands, unit E8M0 block scales, per-row / "
                             "per-output-channel scales in the epilogue, fp32 accumulate; packed weights straight from global memory where the shape allows)"}[prec]
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": measured_traffic(single, prec, B, T),
                "algorithmic_mb_per_launch": round(g_by / g_n / 1e6, 1),
                "launches_per_step": g_n // args.profile_steps, "avg_launch_us": round(g_ms * 1e3 / g_n, 2),
                "gflop_per_launch": round(g_fl / g_n / 1e9, 3), "gemm_ms_per_step": round(g_ms / args.profile_steps, 3),
                "attention": {"achieved": round(a_fl / (a_ms * 1e-3) / 1e12, 2), "ms_per_step": round(a_ms / args.profile_steps, 3),
                              "launches_per_step": a_n // args.profile_steps}}
        small = small_launch_note(single, B, T)
        if small:
            roof["note"] = small
        if other:
            roof["bf16_launches"] = other
            roof.update(blended)
        if fp32_side:
            roof["fp32_launches"] = fp32_side
        try:
            roof["clock"] = None if (args.no_clock or not with_clock) else loop_clock(prec, 4 * B * T if not single else 2 * B * T, peak, ach)
        except Exception as e:          # a diagnostic next to the measurement, never a reason to lose the line
            roof["clock"] = {"error": repr(e)}
        return roof

    roof = live_roofline(smp, args.precision) if (rank == 0 and args.profile_steps > 0) else None

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
        alt_smp.set_schedule(args.sampler)
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
        alt = {"mode": "fp32_split (two-way fp16 operand split, three fp16 MFMAs per product block, fp32 accumulate)", "ms_per_step": round(a_ms, 3),
               "value": round(B / (a_ms * 1e-3 * S), 5), "unit": "motions/s", "rel_rms_vs_fp32_after_2_steps": rel,
               "achieved_tflops_algorithmic": round(algorithmic_flops_per_motion_step(T, single=single) * B / (a_ms * 1e-3) / 1e12, 2)}
        # its own roofline: live HIP-event pairs around every GEMM launch of an eager pass, against 2500 / 3 algorithmic TFLOP/s
        if args.profile_steps > 0:
            alt_smp.begin(cond, xT)
            alt_smp.profile(True)
            alt_smp.run(min(args.profile_steps, S), use_graph=False)
            g_ms, g_n, g_fl, g_by = alt_smp.profile_read(0)
            a2_ms, a2_n, a2_fl, _ = alt_smp.profile_read(1)
            q_ms, q_n, q_fl, _ = alt_smp.profile_read(3)
            alt_smp.profile(False)
            pk = round(PEAK_BF16_MFMA_TFLOPS / 3, 1)
            ach2 = g_fl / (g_ms * 1e-3) / 1e12
            alt["roofline"] = {"bound": "mfma", "kernel": "gemm_splitw_kernel (fp32 result from 3 x v_mfma_f32_32x32x16_f16 on two-way fp16 operand splits; packed weights straight from global memory, 128x128 tiles, two workgroups per CU)",
                               "achieved": round(ach2, 2), "peak": pk, "unit": "TFLOP/s", "frac": round(ach2 / pk, 4), "traffic": measured_traffic(single, "fp32_split", B, T),
                               "algorithmic_mb_per_launch": round(g_by / g_n / 1e6, 1), "launches_per_step": g_n // args.profile_steps, "avg_launch_us": round(g_ms * 1e3 / g_n, 2),
                               "gemm_ms_per_step": round(g_ms / args.profile_steps, 3),
                               "attention": {"achieved": round(a2_fl / (a2_ms * 1e-3) / 1e12, 2), "ms_per_step": round(a2_ms / args.profile_steps, 3), "launches_per_step": a2_n // args.profile_steps}}
            if small_launch_note(single, B, T):
                alt["roofline"]["note"] = small_launch_note(single, B, T)
            if q_n:
                alt["roofline"]["fp32_launches"] = {"achieved": round(q_fl / (q_ms * 1e-3) / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "launches_per_step": q_n // args.profile_steps,
                                                    "ms_per_step": round(q_ms / args.profile_steps, 3)}
            try:
                alt["roofline"]["clock"] = None if args.no_clock else loop_clock("fp32_split", 4 * B * T, pk, ach2)
            except Exception as e:
                alt["roofline"]["clock"] = {"error": repr(e)}
        alt_smp.close()

    # configs[4]'s arithmetic (bf16 path, fp8 e4m3 QKV / FFN GEMM operands) on the SAME workload and inputs, under the same clock as the headline:
    # a throughput configuration with a stated accuracy, reported beside the headline, never as it.  Its `roofline` prices
    # the fp8 launches against the 5 PFLOP/s dense fp8 peak and the bf16 launches against 2.5 (live HIP-event pairs, as above).
    fp8 = None
    if world == 1 and args.precision == "fp32" and not args.no_side and not single and not args.facade:
        f_smp = Sampler(d_heads=8, m_heads=8, max_batch=B, max_frames=T, precision="bf16_fp8", **FULL_DIMS)
        f_smp.load_state_dict(sd_cpu)
        f_smp.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
        f_smp.prepare()
        f_smp.set_schedule(args.sampler)
        f_smp.begin(cond, xT)
        f_smp.run(max(args.warmup, 5), use_graph)
        torch.cuda.synchronize()
        nst = max(args.steps, 20)
        t1 = time.perf_counter()
        f_smp.run(nst, use_graph)
        torch.cuda.synchronize()
        f_ms = (time.perf_counter() - t1) / nst * 1e3
        fp8 = {"mode": "bf16_fp8 (BASELINE configs[4]: bf16 path, QKV / cross-attention input / FFN GEMMs on fp8 e4m3 operands through v_mfma_scale_f32_32x32x64_f8f6f4; "
                       "residual stream, softmax statistics, geometry, blend and DDIM fp32) -- stated accuracy, not a parity claim",
               "steps": nst, "ms_per_step": round(f_ms, 3), "value": round(B / (f_ms * 1e-3 * S), 5), "unit": "motions/s",
               "outputs_finite": bool(torch.isfinite(f_smp.state()["x"]).all().item()),
               "achieved_tflops_algorithmic": round(algorithmic_flops_per_motion_step(T) * B / (f_ms * 1e-3) / 1e12, 2)}
        if args.profile_steps > 0:
            f_smp.begin(cond, xT)
            fp8["roofline"] = live_roofline(f_smp, "bf16_fp8", with_clock=False)
        f_smp.close()

    # The reference's evaluation caller (src/evaluation/datasets.py:100-116: one forward_test per item, the item's own length) on a few items:
    # the sequential loop as shipped against ragged batches, bit-identity asserted in the object (bench.py --eval-items N is the full line)
    ev = None
    if world == 1 and args.precision == "fp32" and not args.no_side and not single and not args.facade and args.sampler == "ddim1000":
        try:
            e = eval_items_bench(args, device, rank, world, as_side=True)
            ev = {k: e[k] for k in ("metric", "value", "unit", "strategies", "bit_identical_to_sequential", "roofline", "graph_cache")}
            ev["items"], ev["sampler"] = e["config"]["items"], e["config"]["sampler"]
        except Exception as ex:          # a side measurement: never a reason to lose the line
            ev = {"error": repr(ex)}

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
        full = {"steps": S, "wall_s": round(wall, 3), "motions_per_s": round(world * B / wall, 5), "seconds_per_motion": round(wall / B, 4),
                "outputs_finite": bool(torch.isfinite(out).all().item())}
        del out

    # The reference's entry points end to end: MixerMDM.forward (src/scripts/infer/mixermdm.py:119-141: every history list of the 50 steps
    # is kept) and forward_test (src/evaluation/datasets.py:101-116), text conditioning precomputed (`cond` in the batch: the CLIP tower is
    # upstream of this path).  Second calls: the (B, T, S) graph and the workspace exist.
    fac = None
    if model is not None and rank == 0:
        batch = {"cond": cond, "x_T": xT, "motion_lens": torch.full((B, 1), T, dtype=torch.long)}
        fac = {}
        for name, fn in (("forward_test", model.forward_test), ("forward", model.forward)):
            fn(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            o = fn(batch)
            torch.cuda.synchronize()
            w = time.perf_counter() - t1
            fac[name] = {"wall_s": round(w, 4), "seconds_per_motion": round(w / B, 4), "outputs_finite": bool(torch.isfinite(o["output"]).all().item()),
                         "history_lists": {k: len(v) for k, v in o.items() if k != "output"}}
            del o

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:       # reported at N=1 only (the other ranks of an N>1 run would idle behind it)
        cpu = cpu_baseline(sd_cpu, stats, T, args.cpu_steps, single, args.sampler, S, B, whole_host=not args.no_whole_host)

    if rank == 0:
        flops = algorithmic_flops_per_motion_step(T, single=single)
        value = world * B / (ms_per_step * 1e-3 * S)
        which = "configs[1]" if single else ("configs[3] (batch 256 over 8 GPUs = 32 per GPU)" if world == 8 and B == 32 else
                                             "configs[3]'s per-GPU shard (32) on %d GPU(s)" % world if B == 32 else "configs[2]" if (B, T, args.sampler) == (16, 300, "ddim1000") else
                                             "the reference's own caller shape (src/scripts/infer/mixermdm.py:73,117-124; src/evaluation/datasets.py:58,100-116)" if args.sampler == "ddim50" else "configs[2] shape family")
        which1 = "configs[1]" if (B, T, args.sampler) == (32, 196, "ddim1000") else "configs[0] (the reference's CPU-runnable case, here on the GPU)" if (B, T, args.sampler) == (1, 120, "ddim50") else "configs[1] shape family"
        wl = ("BASELINE %s: single-person in2IN (individual denoiser, CFG 3.5), T=%d, %s (eta=0), batch %d per GPU, %s, random-init weights" % (which1, T, args.sampler, B, args.precision)) if single else \
             ("BASELINE %s: 2-person MixerMDM (in2IN individual + in2IN interaction + Mixer mode 4, align, CFG 3.5), "
              "T=%d, %s (eta=0), batch %d per GPU, %s, random-init weights" % (which, T, args.sampler, B, {"fp32": "fp32", "fp32_split": "fp32 via two-way fp16 operand split (three fp16 MFMAs per product block)", "bf16": "bf16 GEMM operands / fp32 accumulate (configs[4]-style)", "bf16_fp8": "configs[4]: bf16 path with fp8 e4m3 QKV / FFN GEMM operands, fp32 accumulate"}[args.precision]))
        line = {
            "metric": "generated motions/sec (1000-step DDPM schedule sampled with DDIM eta=0%s, T=%d, %s)" % ("" if S == 1000 else " on %d respaced steps (%s)" % (S, args.sampler), T, "single-person" if single else "2-person"),
            "value": round(value, 5), "unit": "motions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_ranks": rank_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "fp32_split": "f32 (2xfp16 operand split, fp32 accumulate)", "bf16": "bf16", "bf16_fp8": "bf16 + fp8 e4m3 QKV/FFN operands"}[args.precision], "data": "synthetic",
            "config": {"workload": wl,
                       "batch_per_gpu": B, "frames": T, "sampler": args.sampler, "sampler_steps": S, "hipgraph": use_graph, "parallelism": "batch-sharded x%d, no in-loop collective" % world},
            "per_gpu_batch": B,
            "achieved_tflops_algorithmic": round(flops * B * world / (ms_per_step * 1e-3) / 1e12, 2),
            "frac_of_f32_mfma_peak": round(flops * B / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if args.precision in ("fp32", "fp32_split") else None,
            "outputs_finite": finite,
            "full_loop": full, "facade": fac, "roofline": roof, "cpu_baseline": cpu, "fp32_split": alt, "bf16_fp8": fp8, "eval_items": ev,
        }
        if world > 1:
            line["n1_same_batch"] = n1_same_batch(B, args.precision)
        line.update(lib_stamp())
        print(json.dumps(line), flush=True)
    if model is None:
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
        # 10 words per workgroup (8 stamps + 2 cycle counters), sized for the smallest tile the dispatch can pick (64 x 64): the stamping
        # kernel writes without a bound check, so the buffer must cover every workgroup of this launch
        buf = torch.zeros(10 * ((M + 63) // 64) * ((N + 63) // 64), dtype=torch.int64, device=d)
        lib.mmdm_diag_set(b"gemm_tail", 0)          # one launch per call: the stamps are indexed by workgroup (the single-chain samplers split the last round off)
        try:
            for _ in range(300):
                call()
            lib.mmdm_diag_set(b"gemm_stamps", buf.data_ptr())
            call(); torch.cuda.synchronize()
        finally:                            # the library's diagnostic switches are process-global: put them back whatever happened
            lib.mmdm_diag_set(b"gemm_stamps", 0)
            lib.mmdm_diag_set(b"gemm_tail", -1)
        kern = lib.mmdm_last_gemm_kernel().decode()
        tm_, tn_ = [int(v) for v in re.search(r"<(\d+),(\d+)", kern).groups()]
        bm, bn = 32 * (tm_ // 10) * (tm_ % 10), 32 * (tn_ // 10) * (tn_ % 10)
        n = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        raw = buf.cpu()
        st, ck = raw[:8 * n].view(n, 8).double(), raw[8 * n:10 * n].view(n, 2).double()
        mhz = (100.0 * (ck[:, 1] - ck[:, 0]) / (st[:, 2] - st[:, 1]).clamp(min=1)).median().item()
    else:
        xs, ws = ops.split_f32(x), ops.split_f32(w)
        wp = ops.split_pack_weight(ws)
        call = lambda: ops.linear_split(xs, wp, b, packed=True)
        for _ in range(300):
            call()
        kern = lib.mmdm_last_gemm_kernel().decode()
        tm_, tn_ = [int(v) for v in re.search(r"<(\d+),(\d+)", kern).groups()]
        bm, bn, waves = 32 * (tm_ // 10) * (tm_ % 10), 32 * (tn_ // 10) * (tn_ % 10), (tm_ // 10) * (tn_ // 10)
        n = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        buf = torch.zeros(n * waves * 8, dtype=torch.int64, device=d)
        try:
            lib.mmdm_diag_set(b"split_timeline", buf.data_ptr())
            call(); torch.cuda.synchronize()
        finally:
            lib.mmdm_diag_set(b"split_timeline", 0)
        t = buf.view(n * waves, 8).double().cpu()
        mhz = (100.0 * t[:, 7] / t[:, 6].clamp(min=1)).median().item()
    pk = peak * mhz / 2400.0
    return {"shader_mhz_in_k_loop": round(mhz), "kernel": kern, "peak_at_that_clock": round(pk, 1), "frac_of_peak_at_that_clock": round(achieved / pk, 4),
            "how": "s_memtime / s_memrealtime between K-loop entry and exit of every workgroup (median) on one %dx%dx%d launch after 300 warm ones; "
                   "`peak` above is the guide's 2.4 GHz figure" % (M, N, K)}


def eval_items_bench(args, device, rank, world, as_side=False):
    """The reference's evaluation caller (src/evaluation/datasets.py:58, 100-116: one forward_test per item, B = 1, the item's own length) on N
    synthetic items, T uniform in [60, 300]: today's sequential loop against items in flight over one weight set and against ragged batches.
    Every strategy produces the same bits per item (checked here on the outputs).  One JSON line; `value` = items/s of the ragged strategy.
    as_side: the `eval_items` side object of the default line -- args.side_eval_items items, fp32, ddim50, the shipped sequential loop against ragged
    batches only, no CPU leg -- returned, not printed."""
    import numpy as np
    import torch
    from mixermdm_amd.configs import get_config
    from mixermdm_amd.models import MixerMDM
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, FULL_DIMS
    if world != 1:
        raise SystemExit("--eval-items measures one GPU (items shard over ranks without any collective: generation.generate_for_evaluation(shard_items=True))")
    sampler = "ddim50" if args.sampler == "ddim1000" else args.sampler
    N = args.side_eval_items if as_side else args.eval_items
    precision = "fp32" if as_side else args.precision
    rng = np.random.RandomState(0)
    lens = [int(v) for v in rng.randint(60, 301, size=N)]
    sd_cpu = synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS)
    stats = synthetic_stats()
    model = MixerMDM(get_config(os.path.join(ROOT, "configs", "models", "MixerMDM.yaml")), num_frames=300, sampling_strategy=sampler, config_root=ROOT)
    model.precision = precision
    model.load_state_dict({"mixing." + k: v for k, v in sd_cpu.items()})
    model.set_norm_stats(stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"])
    model = model.to(device).eval()
    batches = []
    for i, T in enumerate(lens):
        g = torch.Generator().manual_seed(100 + i)
        batches.append({"cond": torch.randn(1, 8 * 768, generator=g).to(device), "x_T": torch.randn(1, T, 524, generator=g).to(device), "motion_lens": torch.tensor([T])})
    S = int(sampler[4:]) if sampler.startswith("ddim") else 1000
    flops = sum(algorithmic_flops_per_motion_step(T) for T in lens) * S
    peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "fp32_split": round(PEAK_BF16_MFMA_TFLOPS / 3, 1), "bf16_fp8": PEAK_FP8_MFMA_TFLOPS}[precision]

    def run(name, fn, warm):
        with torch.no_grad():
            fn(batches[:warm])                       # untimed: library / allocator warm-up, the first graph captures
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn(batches)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        outs = [r["output"] for r in res]
        tf = flops / dt / 1e12
        return outs, {"wall_s": round(dt, 3), "items_per_s": round(N / dt, 4), "seconds_per_item": round(dt / N, 4), "achieved_tflops_algorithmic": round(tf, 2),
                      "frac_of_peak": round(tf / peak, 4), "outputs_finite": bool(all(torch.isfinite(o).all().item() for o in outs))}

    strat = {}
    ref, strat["sequential_forward_test"] = run("seq", lambda b: [model.forward_test(dict(x)) for x in b], 3)
    strat["sequential_forward_test"]["what"] = "the reference's loop as shipped in round 4: model.forward_test(batch) per item (B = 1, influence histories kept as the reference does), one handle, one stream"
    variants = [("sequential", dict(batching="sequential"), "the same loop without the influence history side outputs (the evaluation harness discards them)"),
                ("inflight2", dict(batching="inflight", inflight=2), "2 handles over ONE weight set (mmdm_create_shared), items dealt round-robin, each on its own stream"),
                ("inflight4", dict(batching="inflight", inflight=4), "4 handles over one weight set"),
                ("ragged", dict(batching="ragged", max_rows=args.eval_max_rows), "items packed in call order into ragged batches of <= %d frames (mmdm_begin_ragged: per-sequence lengths as device data)" % args.eval_max_rows)]
    if as_side:
        variants = [v for v in variants if v[0] == "ragged"]
    same = {}
    for name, kw, what in variants:
        outs, strat[name] = run(name, lambda b, kw=kw: model.sample_many([dict(x) for x in b], mode="eval_intermediate", keep_history=False, **kw), 6)
        strat[name]["what"] = what
        same[name] = bool(all(torch.equal(a, b) for a, b in zip(outs, ref)))
        strat[name]["speedup_vs_sequential_forward_test"] = round(strat["sequential_forward_test"]["wall_s"] / strat[name]["wall_s"], 3)
    # dominant kernel of the ragged strategy: live HIP-event pairs around every GEMM launch of two eager steps of the first ragged batch
    roof = None
    if args.profile_steps > 0:
        rows, grp = 0, []
        for i, T in enumerate(lens):
            if grp and rows + T > args.eval_max_rows:
                break
            grp.append(i); rows += T
        smp = model._sampler_for(len(grp), 300)
        smp.begin_ragged(torch.cat([batches[i]["cond"] for i in grp], 0), [batches[i]["x_T"][0] for i in grp], [lens[i] for i in grp])
        smp.profile(True)
        smp.run(args.profile_steps, use_graph=False)
        g_ms, g_n, g_fl, g_by = smp.profile_read(0)
        a_ms, a_n, a_fl, _ = smp.profile_read(1)
        smp.profile(False)
        ach = g_fl / (g_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "the precision mode's GEMM kernel (see the default line) on the first ragged batch: %d items, %d frames in a group of %d rows" % (len(grp), rows, smp.rows),
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": ragged_traffic(precision, smp.rows),
                "launches_per_step": g_n // args.profile_steps, "avg_launch_us": round(g_ms * 1e3 / g_n, 2), "gemm_ms_per_step": round(g_ms / args.profile_steps, 3),
                "attention": {"achieved": round(a_fl / (a_ms * 1e-3) / 1e12, 2), "ms_per_step": round(a_ms / args.profile_steps, 3), "launches_per_step": a_n // args.profile_steps}}
    cpu = None
    if not args.no_cpu_baseline and not as_side:
        # the oracle on ONE item of the median length, a few steps after the thread calibration, extrapolated to the S steps of an item
        Tm = int(np.median(lens))
        c = cpu_baseline(sd_cpu, stats, Tm, args.cpu_steps, False, sampler, S, 1, whole_host=False)
        cpu = {"value": c["value"], "unit": "items/s", "cores": c.get("cores"), "kind": c.get("kind", "port"),
               "sample": "one item of the median length T = %d on the host cores: " % Tm + c.get("sample", ""), "seconds_per_item": c.get("seconds_per_motion")}
    best = strat["ragged"]
    line = {"metric": "evaluation items/s (one forward_test per item, B = 1, %s, T uniform in [60, 300]: src/evaluation/datasets.py:100-116)" % sampler,
            "value": best["items_per_s"], "unit": "items/s", "n_gpus": 1, "steps": N, "warmup": 6, "ms_per_step": round(best["wall_s"] / N * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "fp32_split": "f32 (2xfp16 operand split, fp32 accumulate)", "bf16": "bf16", "bf16_fp8": "bf16 + fp8 e4m3 QKV/FFN operands"}[precision],
            "data": "synthetic",
            "config": {"workload": "the reference's evaluation caller: %d items, T uniform in [60, 300] (mean %.0f), %s, 2-person MixerMDM through mixermdm_amd.models.MixerMDM; a 'step' of this line is one item" % (N, float(np.mean(lens)), sampler),
                       "items": N, "sampler": sampler, "precision": precision, "max_rows": args.eval_max_rows},
            "strategies": strat, "bit_identical_to_sequential": same, "roofline": roof, "cpu_baseline": cpu,
            "graph_cache": dict(zip(("captures", "replays", "cached"), model._sampler.graph_stats()))}
    line.update(lib_stamp())
    if as_side:
        model._sampler.close()
        return line
    print(json.dumps(line), flush=True)


def small_launch_note(single, B, T):
    """A line for `roofline` when the call is too small to fill the machine (nit 7): at the reference's B = 1 call the GEMMs have
    M = 4 T rows -- fewer 128 x 128 tiles at N = 1024 than the chip has CUs -- so the library runs smaller tiles and a launch is ONE tile's K loop
    long: `frac` then reads launch fill and latency, not the kernel's quality."""
    rows = (2 if single else 4) * B * T
    tiles = -(-rows // 128) * 8
    if tiles >= 256:
        return None
    return ("M = %d rows: %d tiles of 128 x 128 at N = 1024 on 256 CUs -- the library halves the tiles (64 x 64 fp32, 64 x 128 split / bf16: gemm_f32.hip, gemm_split.hip) and a launch is "
            "one tile's K loop long: `frac` reads launch fill and latency here, not kernel quality" % (rows, tiles))


def measured_traffic(single, precision="fp32", B=16, T=300):
    """HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes of this workload and precision mode (profiles/gemm_traffic.json
    for fp32, profiles/gemm_traffic_<mode>.json otherwise; written by tools/pmc_summary.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams on gfx950).  bench.py cannot collect PMC counters
    itself.  The artefact records the hash of the kernel sources AND the workload (motions per GPU, frames) it was measured on: null when it
    is absent, was taken at another batch / length (bytes per launch scale with M = rows of the batch), or on other kernel sources than the
    ones this run executes."""
    from mixermdm_amd.build import sources_sha
    # the headline workload's file, or -- for another (batch, frames) -- the file of a PMC pass taken at exactly that workload
    # (profiles/gemm_traffic[_<mode>]_b<B>t<T>.json: tools/profile_b1.sh writes the reference's B = 1, T = 299 call shape)
    stem = "gemm_traffic" if precision == "fp32" else "gemm_traffic_%s" % precision
    path = os.path.join(ROOT, "profiles", stem + ".json")
    alt = os.path.join(ROOT, "profiles", "%s_b%dt%d.json" % (stem, B, T))
    if os.path.exists(alt):
        path = alt
    if single or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if (rec.get("batch", 16), rec.get("frames", 300)) != (B, T):
        return None
    if rec.get("kernel_sources_sha") != sources_sha(precision):
        print("bench.py: %s was measured on other kernel sources (%s != %s): roofline.traffic = null" % (os.path.basename(path), rec.get("kernel_sources_sha"), sources_sha(precision)), file=sys.stderr)
        return None
    return rec["traffic_bytes_per_launch"]


def ragged_traffic(precision, rows):
    """HBM-side bytes per GEMM launch of a ragged batch from the committed PMC pass of that workload (profiles/gemm_traffic_ragged.json: one ragged
    batch of `rows` frame rows, fp32; tools/profile_ragged.sh) -- under the same conditions as measured_traffic: same kernel sources, same row
    bucket (bytes per launch scale with the rows of the group)."""
    from mixermdm_amd.build import sources_sha
    path = os.path.join(ROOT, "profiles", "gemm_traffic_ragged.json")
    if precision != "fp32" or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("rows") != rows or rec.get("kernel_sources_sha") != sources_sha("fp32"):
        return None
    return rec["traffic_bytes_per_launch"]


def _oracle_mixer(sd_cpu, stats):
    from oracle import mixer as MX
    from oracle.layers import pe_table
    W = dict(sd_cpu)
    W["sequence_pos_encoder.pe"] = pe_table(512)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    W["denoiser2.sequence_pos_encoder.pe"] = pe_table(1024)
    return W, (stats["mean_hml"], stats["std_hml"], stats["mean_ih"], stats["std_ih"]), MX.MixerSpec(d_heads=8, m_heads=8)


def cpu_baseline(sd_cpu, stats, T, nsteps, single=False, strategy="ddim1000", S=1000, B_gpu=16, whole_host=True):
    """The oracle (a PyTorch-CPU port of the reference path, parity-pinned by tests/golden) timed on this host's cores at B=1 on the SAME
    workload (frames, sampling strategy): a short schedule (ddim50: the reference's own callers) is run IN FULL, the 1000-step schedule is
    sampled -- `nsteps` consecutive DDIM steps after the thread-count calibration -- and extrapolated.  `whole_host`: the host's motions/s
    when it runs k = host_threads / 32 such processes side by side (one process uses 16 of the box's threads)."""
    import torch
    from oracle import mixer as MX, schedule as OS
    from mixermdm_amd.synthetic import synthetic_inputs
    if single:
        return cpu_baseline_single(sd_cpu, T, nsteps, strategy, S)
    W, ostats, spec = _oracle_mixer(sd_cpu, stats)
    sch = OS.make_schedule("cosine", 1000, strategy)
    cond, xT = synthetic_inputs(1, T)
    x, x2 = xT.clone(), xT.clone()
    threads0 = torch.get_num_threads()
    with torch.no_grad():
        # thread-count calibration (one untimed step each): B=1 GEMMs are small, all cores is not always the fastest
        ncpu = os.cpu_count() or 1
        best, calib = (None, 1e30), {}
        # (all hardware threads is never the fastest for B=1 and can be pathological -- 265 s per step on a 256-thread host -- so the
        # calibration stops at 64; `host_threads` reports what the box has)
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            t0 = time.perf_counter()
            MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 1, x, x2, cond)
            dt1 = time.perf_counter() - t0
            calib[nt] = dt1
            if dt1 < best[1]:
                best = (nt, dt1)
        torch.set_num_threads(best[0])
        full = S * best[1] <= 45.0                    # the whole loop fits the bench's time box: time THE workload, not a sample of it
        if full:
            t0 = time.perf_counter()
            for i in range(S - 1, -1, -1):
                x, x2, _, p2 = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, i, x, x2, cond)
            wall = time.perf_counter() - t0
            dt, n_timed = wall / S, S
        else:
            x, x2, _, _ = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 1, x, x2, cond)
            t0 = time.perf_counter()
            for k in range(nsteps):
                x, x2, _, _ = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 2 - k, x, x2, cond)
            dt, n_timed = (time.perf_counter() - t0) / nsteps, nsteps
        # SURVEY 8d also asks the GPU's own batch on the CPU: one untimed step (first touch, allocations), then one timed step at B = 16 on 32
        # threads (64 measured slower: 25.4 vs 15.9 s), extrapolated like the B = 1 figure; it says how much batch efficiency the CPU gets
        bN = None
        if nsteps >= 4 and not full and B_gpu >= 16:
            c16, x16 = synthetic_inputs(16, T)
            nt16 = min(ncpu, 32)
            torch.set_num_threads(nt16)
            MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 1, x16, x16, c16)
            t0 = time.perf_counter()
            MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 2, x16, x16, c16)
            d16 = time.perf_counter() - t0
            bN = {"s_per_step": round(d16, 3), "cores": nt16, "motions_per_s": round(16.0 / (d16 * S), 7),
                  "sample": "the second of two DDIM steps at B=16 (the GPU's batch) on %d threads, extrapolated x%d steps" % (nt16, S)}
        torch.set_num_threads(best[0])
    out = {"value": round(1.0 / (dt * S), 7), "unit": "motions/s", "cores": best[0], "host_threads": ncpu, "kind": "port",
           "sample": ("the WHOLE %s loop (%d DDIM steps) of the same workload at B=1, T=%d on the host CPU (PyTorch %s, fp32): %.2f s, %.3f s/step" % (strategy, S, T, torch.__version__, dt * S, dt)
                      if full else "%d consecutive DDIM steps (i=%d..) of the same workload at B=1, T=%d on the host CPU (PyTorch %s, fp32), %.3f s/step, extrapolated x%d steps"
                      % (n_timed, S - 2, T, torch.__version__, dt, S)) +
                     "; thread count calibrated over {8, 16, 32, 64} of the host's %d hardware threads: %d fastest (one step: %s)"
                     % (ncpu, best[0], ", ".join("%d thr %.2f s" % kv for kv in sorted(calib.items()))),
           "s_per_step_b1": round(dt, 4), "seconds_per_motion": round(dt * S, 3), "extrapolated": not full, "b16": bN}
    if whole_host:
        try:
            out["whole_host"] = cpu_whole_host(T, ncpu, S, strategy)
        except Exception as e:          # a side measurement: never a reason to lose the line
            out["whole_host"] = {"error": repr(e)}
    torch.set_num_threads(threads0)
    return out


def cpu_whole_host(T, ncpu, S, strategy, threads=16, steps=2):
    """k = host_threads / 32 concurrent oracle processes of `threads` threads each (one B=1 chain per process, as a user of the reference
    would fill the host): every worker draws the same synthetic weights, runs one untimed step, waits for the others, then times `steps`
    DDIM steps.  Host motions/s = sum over workers of 1 / (S x its s/step)."""
    import tempfile
    k = max(1, ncpu // 32)
    d = tempfile.mkdtemp(prefix="mmdm_cpu_")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads), str(steps), str(T), os.path.join(d, "w%d" % i)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS=str(threads), MMDM_CPU_STRATEGY=strategy,
                                                                                           HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")) for i in range(k)]
    t0 = time.perf_counter()
    try:
        while not all(os.path.exists(os.path.join(d, "w%d.ready" % i)) for i in range(k)):
            if any(p.poll() not in (None, 0) for p in procs) or time.perf_counter() - t0 > 240:
                raise RuntimeError("a cpu worker failed or did not get ready in 240 s")
            time.sleep(0.2)
        open(os.path.join(d, "go"), "w").close()
        for p in procs:
            p.wait(timeout=240)
        per = [json.load(open(os.path.join(d, "w%d.json" % i)))["s_per_step"] for i in range(k)]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    return {"processes": k, "threads_each": threads, "s_per_step": [round(v, 3) for v in per], "motions_per_s": round(sum(1.0 / (v * S) for v in per), 7),
            "sample": "%d concurrent B=1 oracle processes x %d threads (= %d of the host's %d hardware threads), %d timed DDIM steps each after one untimed step and a "
                      "start barrier, extrapolated x%d steps" % (k, threads, k * threads, ncpu, steps, S)}


def cpu_worker(threads, steps, frames, path):
    """One process of cpu_whole_host (hidden --cpu-worker mode; never touches a GPU)."""
    import torch
    from oracle import mixer as MX, schedule as OS
    from mixermdm_amd.synthetic import synthetic_state_dict, synthetic_stats, synthetic_inputs, FULL_DIMS
    threads, steps, T = int(threads), int(steps), int(frames)
    torch.set_num_threads(threads)
    W, ostats, spec = _oracle_mixer(synthetic_state_dict(seed=0, std=0.02, bias_std=0.0, **FULL_DIMS), synthetic_stats())
    sch = OS.make_schedule("cosine", 1000, os.environ.get("MMDM_CPU_STRATEGY", "ddim1000"))
    S = sch.num_timesteps
    cond, xT = synthetic_inputs(1, T)
    with torch.no_grad():
        x, x2, _, _ = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 1, xT, xT.clone(), cond)
        open(path + ".ready", "w").close()
        go = os.path.join(os.path.dirname(path), "go")
        t0 = time.perf_counter()
        while not os.path.exists(go):
            if time.perf_counter() - t0 > 240:
                raise SystemExit(1)
            time.sleep(0.05)
        t0 = time.perf_counter()
        for k in range(steps):
            x, x2, _, _ = MX.mixer_ddim_step(W, spec, ostats, sch, 3.5, S - 2 - k, x, x2, cond)
        dt = (time.perf_counter() - t0) / steps
    with open(path + ".json", "w") as f:
        json.dump({"s_per_step": dt, "threads": threads}, f)


def cpu_baseline_single(sd_cpu, T, nsteps, strategy="ddim1000", S=1000):
    """configs[0] / configs[1] on the CPU: the single-person in2IN sampler at B=1.  A short schedule (configs[0]: T=120, ddim50) is timed in
    full, the 1000-step one is sampled and extrapolated."""
    import torch
    from oracle import mixer as MX, schedule as OS
    from oracle.layers import pe_table
    from mixermdm_amd.synthetic import synthetic_inputs
    W = dict(sd_cpu)
    W["denoiser1.sequence_pos_encoder.pe"] = pe_table(1024)
    sch = OS.make_schedule("cosine", 1000, strategy)
    cond, x = synthetic_inputs(1, T, single=True)
    step = lambda i, x: MX.ddim_update(sch, i, x, MX.cfg_single(W, "denoiser1.", "individual", 3.5, x, torch.full((1,), sch.timestep_map[i], dtype=torch.long), cond, 8))
    with torch.no_grad():
        t0 = time.perf_counter()
        x = step(S - 1, x)
        d0 = time.perf_counter() - t0
        full = S * d0 <= 45.0
        t0 = time.perf_counter()
        if full:
            x = synthetic_inputs(1, T, single=True)[1]
            for i in range(S - 1, -1, -1):
                x = step(i, x)
            n_timed = S
        else:
            for k in range(nsteps):
                x = step(S - 2 - k, x)
            n_timed = nsteps
        dt = (time.perf_counter() - t0) / n_timed
    return {"value": round(1.0 / (dt * S), 7), "unit": "motions/s", "cores": torch.get_num_threads(), "host_threads": os.cpu_count(), "kind": "port",
            "sample": ("the WHOLE %s loop (%d DDIM steps) of the single-person workload at B=1, T=%d on the host CPU (PyTorch %s, fp32): %.2f s" % (strategy, S, T, torch.__version__, dt * S)) if full else
                      ("%d consecutive DDIM steps of the single-person workload at B=1, T=%d on the host CPU (PyTorch %s, fp32), %.3f s/step, extrapolated x%d steps" % (nsteps, T, torch.__version__, dt, S)),
            "s_per_step_b1": round(dt, 4), "seconds_per_motion": round(dt * S, 3), "extrapolated": not full}


if __name__ == "__main__":
    main()
