"""Summary of tools/pmc_fp8.sh: per fp8 GEMM kernel (and grid) the averages of every collected counter, plus derived figures."""
import sys, glob, csv, collections, json, os
out = sys.argv[1]
dur, cnt, n = {}, collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(lambda: collections.Counter())
for p in sorted(glob.glob(out + "/pass*")):
    if not os.path.isdir(p):
        continue
    d = {}
    for f in glob.glob(p + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_bf16" not in k and "gemm_fp8p" not in k:
                continue
            key = (k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:], r["Grid_Size"])
            c = r["Counter_Name"]
            cnt[key][c] += float(r["Counter_Value"]); n[key][c] += 1
            if c == "GRBM_GUI_ACTIVE":
                cnt[key]["ns:" + os.path.basename(p)] += d.get(r["Dispatch_Id"], 0); n[key]["ns:" + os.path.basename(p)] += 1
res = {}
for key, c in cnt.items():
    a = {k: v / n[key][k] for k, v in c.items()}
    ns = [v for k, v in a.items() if k.startswith("ns:")]
    us = sum(ns) / len(ns) / 1e3
    e = {"kernel": key[0], "grid": key[1], "launches_seen": int(max(n[key].values())), "avg_us_under_pmc": round(us, 1)}
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8          # per XCD
    if cyc:
        e["clock_ghz"] = round(cyc / (us * 1e3), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and cyc:
        e["mfma_busy"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        sq = a["SQ_BUSY_CYCLES"] / 32                  # per shader engine: the SQ's own busy cycles, shader clock domain (GRBM_GUI_ACTIVE reads high on short dispatches)
        e["clock_ghz_sq"] = round(sq / (us * 1e3), 3)
        e["mfma_busy_sq"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * sq), 4)
    if "SQ_WAVE_CYCLES" in a:
        wc = a["SQ_WAVE_CYCLES"]
        e["wave_cycles"] = {k: round(a[k] / wc, 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in a}
    if "TCP_TCC_READ_REQ_sum" in a:
        e["l1_to_l2_read_requests"] = a["TCP_TCC_READ_REQ_sum"]
        e["l1_to_l2_read_TBps_at_64B"] = round(a["TCP_TCC_READ_REQ_sum"] * 64 / (us * 1e-6) / 1e12, 2)
        e["l1_to_l2_read_TBps_at_128B"] = round(a["TCP_TCC_READ_REQ_sum"] * 128 / (us * 1e-6) / 1e12, 2)
    if "TCC_REQ_sum" in a:
        e["l2_requests"] = a["TCC_REQ_sum"]
        e["l2_req_TBps_at_128B"] = round(a["TCC_REQ_sum"] * 128 / (us * 1e-6) / 1e12, 2)
    if "TCC_HIT_sum" in a:
        e["l2_hit"] = round(a["TCC_HIT_sum"] / max(1.0, a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
    if "SQ_LDS_IDX_ACTIVE" in a and cyc:
        e["lds_array_busy_of_cu_cycles"] = round(a["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 4)        # summed over 256 CUs, in CU cycles (quad-cycle units are uncalibrated: ratio between variants)
        e["lds_bank_conflict_share"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, a["SQ_LDS_IDX_ACTIVE"]), 4)
    for k in ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_INSTS_VALU_MFMA_MOPS_F8"):
        if k in a:
            e[k] = a[k]
    if "FETCH_SIZE" in a:
        e["hbm_side_read_mb"] = round(2 * a["FETCH_SIZE"] / 1024, 1)      # FETCH_SIZE is in KiB; x2: the guide's gfx950 correction for 16-byte-per-lane streams
    if "WRITE_SIZE" in a:
        e["hbm_side_write_mb"] = round(a["WRITE_SIZE"] / 1024, 1)
    res["%s grid %s" % key] = e
print(json.dumps(res, indent=1))
