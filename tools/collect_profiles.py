"""Copy what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into profiles/ (tracked) and print the figures DESIGN.md quotes.
usage: python tools/collect_profiles.py [round-tag, default r04]"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")
for n in ("serial", "overlap", "fp32_split_serial", "bf16_serial", "bf16_fp8_serial", "single_serial"):
    st = os.path.join(src, "summary", f"kernel_stats_{n}.csv")
    if not os.path.exists(st):
        continue
    shutil.copy(st, os.path.join(dst, f"{tag}_kernel_stats_{n}.csv"))
    for line in open(os.path.join(src, f"{n}.json")):
        if line.startswith("{"):
            d = json.loads(line)
            json.dump(d, open(os.path.join(dst, f"{tag}_bench_{n}_under_rocprof.json"), "w"), indent=1)
            r = d["roofline"]
            print(f"{n:18s} {d['ms_per_step']:7.3f} ms/step  GEMM live {r['achieved']:7.2f} TFLOP/s (frac {r['frac']:.4f}, {r['avg_launch_us']:.1f} us/launch)  attention {r['attention']}")
for p in ("fp32", "fp32_split", "bf16_fp8"):
    sm = os.path.join(src, f"summary_{p}")
    if not os.path.exists(os.path.join(sm, "pmc_summary.json")):
        continue
    shutil.copy(os.path.join(sm, "pmc_summary.json"), os.path.join(dst, f"{tag}_pmc_{'summary' if p == 'fp32' else p.replace('bf16_fp8', 'fp8')}.json"))
    if os.path.exists(os.path.join(sm, "gemm_traffic.json")):
        out = os.path.join(dst, "gemm_traffic.json" if p == "fp32" else f"gemm_traffic_{p}.json")
        shutil.copy(os.path.join(sm, "gemm_traffic.json"), out)
        t = json.load(open(out))
        print(f"traffic {p:11s} sha {t['kernel_sources_sha']}  {t['traffic_bytes_per_launch'] / 1e6:.0f} MB/launch  L2 hit {t['l2_hit_rate']}  MFMA busy {t.get('mfma_busy_frac')} (sq {t.get('mfma_busy_frac_sq')}, sq clock {t.get('clock_ghz_sq')})  "
              f"clock {t.get('clock_ghz')} GHz  {t.get('hbm_side_TBps')} TB/s beyond L2")
# the reference's own call shape (tools/profile_b1.sh): B = 1, T = 299, ddim50 through the facade
b1 = os.path.join(ROOT, "gpurun_out", f"prof_b1_{tag}")
for n in ("serial", "overlap"):
    st = os.path.join(b1, "summary", f"kernel_stats_{n}.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(dst, f"{tag}_kernel_stats_infer_b1_{n}.csv"))
        for line in open(os.path.join(b1, f"{n}.json")):
            if line.startswith("{"):
                d = json.loads(line)
                json.dump(d, open(os.path.join(dst, f"{tag}_bench_infer_b1_{n}_under_rocprof.json"), "w"), indent=1)
                print(f"infer_b1 {n:8s} {d['ms_per_step']:7.3f} ms/step  GEMM live frac {d['roofline']['frac']}  facade {d.get('facade')}")
if os.path.exists(os.path.join(b1, "summary_fp32", "pmc_summary.json")):
    shutil.copy(os.path.join(b1, "summary_fp32", "pmc_summary.json"), os.path.join(dst, f"{tag}_pmc_infer_b1.json"))
    if os.path.exists(os.path.join(b1, "summary_fp32", "gemm_traffic.json")):
        shutil.copy(os.path.join(b1, "summary_fp32", "gemm_traffic.json"), os.path.join(dst, "gemm_traffic_b1t299.json"))
# configs[4]'s per-GPU shard (tools/profile_fp8_b64.sh): bf16_fp8 at B = 64
f8 = os.path.join(ROOT, "gpurun_out", f"prof_fp8_b64_{tag}")
if os.path.exists(os.path.join(f8, "summary", "kernel_stats_serial.csv")):
    shutil.copy(os.path.join(f8, "summary", "kernel_stats_serial.csv"), os.path.join(dst, f"{tag}_kernel_stats_fp8_b64_serial.csv"))
    for line in open(os.path.join(f8, "serial.json")):
        if line.startswith("{"):
            d = json.loads(line)
            json.dump(d, open(os.path.join(dst, f"{tag}_bench_fp8_b64_serial_under_rocprof.json"), "w"), indent=1)
            print(f"fp8_b64 serial {d['ms_per_step']:7.3f} ms/step  fp8 GEMM live frac {d['roofline']['frac']}")
    if os.path.exists(os.path.join(f8, "summary_bf16_fp8", "pmc_summary.json")):
        shutil.copy(os.path.join(f8, "summary_bf16_fp8", "pmc_summary.json"), os.path.join(dst, f"{tag}_pmc_fp8_b64.json"))
    if os.path.exists(os.path.join(f8, "summary_bf16_fp8", "gemm_traffic.json")):
        shutil.copy(os.path.join(f8, "summary_bf16_fp8", "gemm_traffic.json"), os.path.join(dst, "gemm_traffic_bf16_fp8_b64t300.json"))
        t = json.load(open(os.path.join(dst, "gemm_traffic_bf16_fp8_b64t300.json")))
        print(f"traffic fp8 B=64 sha {t['kernel_sources_sha']}  {t['traffic_bytes_per_launch'] / 1e6:.0f} MB/launch  L2 hit {t['l2_hit_rate']}  MFMA busy {t.get('mfma_busy_frac')} (sq {t.get('mfma_busy_frac_sq')})  {t.get('hbm_side_TBps')} TB/s beyond L2")
# one ragged batch of the evaluation caller (tools/profile_ragged.sh)
rg = os.path.join(ROOT, "gpurun_out", f"prof_ragged_{tag}")
if os.path.exists(os.path.join(rg, "summary", "kernel_stats_serial.csv")):
    shutil.copy(os.path.join(rg, "summary", "kernel_stats_serial.csv"), os.path.join(dst, f"{tag}_kernel_stats_ragged.csv"))
    if os.path.exists(os.path.join(rg, "summary_pmc", "pmc_summary.json")):
        shutil.copy(os.path.join(rg, "summary_pmc", "pmc_summary.json"), os.path.join(dst, f"{tag}_pmc_ragged.json"))
    if os.path.exists(os.path.join(rg, "summary_pmc", "gemm_traffic.json")):
        shutil.copy(os.path.join(rg, "summary_pmc", "gemm_traffic.json"), os.path.join(dst, "gemm_traffic_ragged.json"))
        t = json.load(open(os.path.join(dst, "gemm_traffic_ragged.json")))
        print(f"traffic ragged ({t.get('rows')} rows) sha {t['kernel_sources_sha']}  {t['traffic_bytes_per_launch'] / 1e6:.0f} MB/launch  L2 hit {t['l2_hit_rate']}  MFMA busy {t.get('mfma_busy_frac')} / sq {t.get('mfma_busy_frac_sq')}")
# the stand-alone fp8 GEMM under the counters (tools/pmc_fp8.sh)
f8 = os.path.join(ROOT, "gpurun_out", f"pmc_fp8_{tag}", "summary.txt")
if os.path.exists(f8):
    shutil.copy(f8, os.path.join(dst, f"{tag}_pmc_fp8_gemm.json"))
rows = list(csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats_serial.csv"))))
g = [r for r in rows if r["Name"].startswith(("gemm_glds_kernel", "gemm_mix_kernel", "gemm_s16_kernel"))]
tot, n = sum(float(r["TotalDurationNs"]) for r in g), sum(int(r["Calls"]) for r in g)
steps = next(int(r["Calls"]) for r in rows if "xstart_ddim_kernel" in r["Name"])
print(f"serial trace: {steps} steps; fp32 GEMM kernels {tot / 1e6:.1f} ms over {n} launches = {tot / n / 1e3:.1f} us average, {tot / steps / 1e6:.2f} ms/step "
      f"= {136 * 51.453e9 / (tot / steps * 1e-9) / 1e12:.1f} TFLOP/s")
for r in g + [r for r in rows if r["Name"].startswith(("attn_", "adaln"))]:
    print(f"  {r['Name'][:60]:60s} x{r['Calls']:>4s}  {float(r['AverageNs']) / 1e3:7.1f} us  {r['Percentage']} %")
