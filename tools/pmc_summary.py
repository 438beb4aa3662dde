"""Summarise rocprofv3 counter-collection CSVs (one directory per --pmc pass) for one kernel.

usage: python tools/pmc_summary.py OUT.json KERNEL_SUBSTRING DIR [DIR ...]
Averages each counter over the kernel's dispatches (first dispatch of each pass dropped as warm-up),
applies the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE
reads half of a wide coalesced stream) and derives the ratios quoted in DESIGN.md."""
import csv, glob, json, os, sys

out, kern, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
vals, dur = {}, []
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if kern not in row["Kernel_Name"]:
                continue
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                dur.append((row["Dispatch_Id"], f, int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        for name, by in per.items():
            ids = sorted(by, key=int)[1:] or sorted(by, key=int)
            vals[name] = sum(by[i] for i in ids) / len(ids)
c = vals
der = {}
if "FETCH_SIZE" in c:
    der["fetch_bytes_per_launch_corrected"] = c["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in c:
    der["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    der["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    der["tcc_miss_bytes"] = c["TCC_MISS_sum"] * 128
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
    # both counters are sums over the 8 XCDs; MFMA busy additionally over the 32 CUs x 4 SIMDs of each XCD
    der["mfma_busy_fraction_of_simd_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c.get("GRBM_GUI_ACTIVE", 0) * 128) if c.get("GRBM_GUI_ACTIVE") else None
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    der["wave_cycles_waiting_fraction"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
if "SQ_WAIT_INST_LDS" in c and "SQ_WAVE_CYCLES" in c:
    der["wave_cycles_waiting_on_lds_fraction"] = c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]
if "SQ_BUSY_CU_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
    der["cu_busy_fraction"] = c["SQ_BUSY_CU_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 32)
if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
    der["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
# launch duration under the counter passes (kernel trace of every pass, first dispatch of each dropped) and, from the
# pass that holds GRBM_GUI_ACTIVE (a sum over the 8 XCDs), the shader clock the kernel actually ran at
durs = []
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        durs += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[1:]]
if durs:
    der["avg_launch_us_under_pmc"] = sum(durs) / len(durs)
    if c.get("GRBM_GUI_ACTIVE"):
        der["shader_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / der["avg_launch_us_under_pmc"] / 1e3
json.dump({"kernel_filter": kern, "counters": c, "derived": der}, open(out, "w"), indent=1)
print(json.dumps(der, indent=1))
