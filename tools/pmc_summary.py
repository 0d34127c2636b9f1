#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass (counter_collection.csv) -> JSON on stdout.

usage: pmc_summary.py <dir with *counter_collection.csv> [n_cu=256]
SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs: mfma_busy_cycles_per_simd = that / (n_cu * 4); divide by the kernel's
duration in cycles (from a --kernel-trace --stats run: GRBM_GUI_ACTIVE is inflated under PMC collection) for the pipe-busy
fraction."""
import csv, glob, json, sys, collections
d = sys.argv[1]; n_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 256
files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
out = {}
for k, c in acc.items():
    n = len(cnt[k]); a = {m: v / n for m, v in c.items()}
    e = {"dispatches": n, "per_dispatch": {m: round(v, 1) for m, v in sorted(a.items())}}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
        e["mfma_busy_cycles_per_simd"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (n_cu * 4), 1)
    if a.get("SQ_WAVE_CYCLES"):
        for m in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if m in a: e[m + "_frac_of_wave_cycles"] = round(a[m] / a["SQ_WAVE_CYCLES"], 3)
    if a.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"], 4)
    out[k[:110]] = e
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["per_dispatch"].get("GRBM_GUI_ACTIVE", 0) * kv[1]["dispatches"])), indent=1))
