"""Summarise a round's rocprofv3 output (tools/profile_round.sh) into the small artefacts committed under profiles/:
   kernel_stats_<pass>.csv (per-kernel totals from the kernel traces), pmc_summary.json (MFMA busy, clock, HBM-side bytes and L2 hit rate
   per kernel) and gemm_traffic.json (what bench.py reports as roofline.traffic for that precision mode, with the hash of the kernel sources
   it was measured on).  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests of 16-byte-per-lane streams at 64 B: MI355X_MICROARCH.md,
   HBM); counters are per-XCD sums as rocprofv3 reports them.
   usage: pmc_summary.py <dir with the pass directories> [<output dir> [<precision: fp32 | fp32_split | bf16_fp8> [traces-only]]]"""
import sys, os, glob, csv, json, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixermdm_amd.build import sources_sha
src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(src, "summary")
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
traces_only = len(sys.argv) > 4
PB, PT = int(os.environ.get("PMC_BATCH", "16")), int(os.environ.get("PMC_FRAMES", "300"))     # the workload the PMC passes were taken at (tools/profile_round.sh: the default)
os.makedirs(dst, exist_ok=True)
# the dominant GEMM kernel(s) of each precision mode (prefix of the demangled name)
DOMINANT = {"fp32": ("gemm_glds_kernel", "gemm_mix_kernel", "gemm_s16_kernel"), "fp32_split": ("gemm_splitw_kernel", "gemm_split_kernel"), "bf16_fp8": ("gemm_bf16w_kernel", "gemm_bf16_kernel"),
            "bf16": ("gemm_bf16w_kernel", "gemm_bf16_kernel")}[prec]


def short(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    return re.sub(r"^void ", "", k).split("(")[0]


def traces(d):
    rows = []
    for f in glob.glob(os.path.join(src, d, "**", "*kernel_trace.csv"), recursive=True):
        rows += [(r["Dispatch_Id"], short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
    return rows


for d in sorted(os.listdir(src)):
    if d.startswith("pmc_") or d.startswith("summary") or not os.path.isdir(os.path.join(src, d)):
        continue
    rows = traces(d)
    if not rows:
        continue
    agg = collections.defaultdict(list)
    for _, k, s, e in rows:
        agg[k].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    with open(os.path.join(dst, f"kernel_stats_{d}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100 * sum(v) / tot, 3), min(v), max(v)])
if traces_only:
    sys.exit(0)

# PMC passes
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(src, "pmc_*")):
    if not os.path.isdir(d):
        continue
    dur = {i: (k, e - s) for i, k, s, e in traces(os.path.basename(d))}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum"):
                per[k]["ns:" + r["Counter_Name"]].append(dur.get(r["Dispatch_Id"], ("", 0))[1])
mean = lambda v: sum(v) / len(v) if v else None
out = {"precision": prec, "kernel_sources_sha": sources_sha(prec), "note": "per-dispatch means over the profiled bench steps (eager launches, one stream); FETCH_SIZE / WRITE_SIZE in KB as "
       "reported, fetch_bytes = 2 x FETCH_SIZE x 1024; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); mfma_busy_frac_sq = the same over SQ_BUSY_CYCLES / 32 shader engines (valid on short dispatches too: clock_ghz above 2.4 marks a dispatch on which the GRBM figure is not)", "kernels": {}}
gemm = {"fetch": [], "write": [], "hit": [], "miss": [], "busy": [], "active": [], "ns": [], "sqbusy": []}
for k, c in per.items():
    e = {"dispatches": len(c.get("GRBM_GUI_ACTIVE", c.get("FETCH_SIZE", c.get("WRITE_SIZE", []))))}
    if c.get("GRBM_GUI_ACTIVE"):
        ns = mean(c["ns:GRBM_GUI_ACTIVE"])
        e["avg_us"] = round(ns / 1e3, 1)
        e["clock_ghz"] = round(mean(c["GRBM_GUI_ACTIVE"]) / 8 / ns, 3)
        e["mfma_busy_frac"] = round(mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * mean(c["GRBM_GUI_ACTIVE"]) / 8), 4)
        if c.get("SQ_BUSY_CYCLES"):
            # the same busy cycles over the SQ's OWN busy cycles (summed over the 32 shader engines): both counters tick in the shader clock domain.
            # GRBM_GUI_ACTIVE / 8 / duration reads high on short dispatches (MI355X_MICROARCH.md, DVFS give-back) -- 2.98 "GHz" on a 70 us fp8
            # GEMM of 3600 workgroups, above the part's 2.4 GHz maximum -- and then understates the busy fraction by the same factor
            sq = mean(c["SQ_BUSY_CYCLES"]) / 32
            e["clock_ghz_sq"] = round(sq / ns, 3)
            e["mfma_busy_frac_sq"] = round(mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * sq), 4)
    if c.get("FETCH_SIZE"):
        e["fetch_bytes"] = round(2 * 1024 * mean(c["FETCH_SIZE"]))
    if c.get("WRITE_SIZE"):
        e["write_bytes"] = round(1024 * mean(c["WRITE_SIZE"]))
    if c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
        e["hbm_side_TBps"] = round((e["fetch_bytes"] + e["write_bytes"]) / mean(c["ns:FETCH_SIZE"]) / 1e3, 3)
    if c.get("TCC_HIT_sum"):
        e["l2_hit_rate"] = round(mean(c["TCC_HIT_sum"]) / (mean(c["TCC_HIT_sum"]) + mean(c["TCC_MISS_sum"])), 4)
    out["kernels"][k] = e
    if k.startswith(DOMINANT):
        for nm, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("hit", "TCC_HIT_sum"), ("miss", "TCC_MISS_sum"), ("busy", "SQ_VALU_MFMA_BUSY_CYCLES"),
                        ("active", "GRBM_GUI_ACTIVE"), ("ns", "ns:GRBM_GUI_ACTIVE"), ("sqbusy", "SQ_BUSY_CYCLES")):
            gemm[nm] += c.get(key, [])
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
if gemm["fetch"] and gemm["write"]:
    fetch, write = 2 * 1024 * mean(gemm["fetch"]), 1024 * mean(gemm["write"])
    rec = {"precision": prec, "kernel": " + ".join(DOMINANT) + " (all instantiations of a step)", "kernel_sources_sha": sources_sha(prec),
           "workload": f"bench.py --precision {prec} --batch {PB} --frames {PT} --steps 2 --warmup 1 --no-graph, MMDM_NO_OVERLAP=1",
           "batch": PB, "frames": PT,        # bench.py reports this file as roofline.traffic only for the same motions per GPU and length
           **({"rows": int(os.environ["PMC_ROWS"]), "workload": "tools/ragged_step.py: one ragged batch of the evaluation caller, %s group rows, 2 eager ddim50 steps, MMDM_NO_OVERLAP=1" % os.environ["PMC_ROWS"]}
              if os.environ.get("PMC_ROWS") else {}),        # a ragged batch (tools/profile_ragged.sh): bench.py --eval-items matches on the group's rows
           "dispatches_averaged": len(gemm["fetch"]), "FETCH_SIZE_KB_per_launch_raw": round(mean(gemm["fetch"]), 1), "WRITE_SIZE_KB_per_launch": round(mean(gemm["write"]), 1),
           "fetch_correction": "x2 (gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16-B-per-lane streams; MI355X_MICROARCH.md, HBM)",
           "traffic_bytes_per_launch": round(fetch + write), "l2_hit_rate": round(mean(gemm["hit"]) / (mean(gemm["hit"]) + mean(gemm["miss"])), 4) if gemm["hit"] else None,
           "mfma_busy_frac": round(sum(gemm["busy"]) / (1024 * sum(gemm["active"]) / 8), 4) if gemm["active"] else None,
           "clock_ghz": round(sum(gemm["active"]) / 8 / sum(gemm["ns"]), 3) if gemm["ns"] else None,
           "mfma_busy_frac_sq": round(sum(gemm["busy"]) / (1024 * sum(gemm["sqbusy"]) / 32), 4) if gemm["sqbusy"] else None,
           "clock_ghz_sq": round(sum(gemm["sqbusy"]) / 32 / sum(gemm["ns"]), 3) if gemm["sqbusy"] and gemm["ns"] else None,
           "hbm_side_TBps": round((fetch + write) / (sum(gemm["ns"]) / len(gemm["ns"])) / 1e3, 3) if gemm["ns"] else None,
           "note": "FETCH_SIZE includes Infinity-Cache hits (operand re-reads that miss the 4 MiB XCD L2)"}
    json.dump(rec, open(os.path.join(dst, "gemm_traffic.json"), "w"), indent=1)
    print(json.dumps(rec))
for k, e in sorted(out["kernels"].items(), key=lambda kv: -(kv[1].get("avg_us") or 0))[:12]:
    print(k[:70], e)
