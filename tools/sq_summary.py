#!/usr/bin/env python3
"""Summarise tools/gpu_sq.sh's rocprofv3 --pmc pass (SQ counters) per kernel into profiles/<tag>_sq_summary.json:
fractions of wave cycles spent waiting / issuing, and the share of LDS-active cycles lost to bank conflicts."""
import collections
import csv
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
tot = collections.defaultdict(collections.Counter)
cnt = collections.Counter()
with open("gpurun_out/pmc_sq/pmc_counter_collection.csv") as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
out = {}
for k, c in tot.items():
    if not any(s in k for s in ("chain", "conv_", "stem", "trp", "bilinear", "crop", "fuse_")):
        continue
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    # matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES counts cycles over all SIMDs (32 per 32x32x16 MFMA); the chip's
    # active time is GRBM_GUI_ACTIVE, which rocprofv3 reports SUMMED over the 8 XCDs (MI355X_MICROARCH.md, DVFS): 1024 SIMDs
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mfma = None if not gui else round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0), 4)
    out[k] = {"launches": cnt[k], "mfma_busy_frac": mfma,
              "mfma_busy_cycles_per_launch": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(cnt[k], 1)),
              "gui_active_per_launch_per_xcd": round(gui / 8.0 / max(cnt[k], 1)),
              "wait_any_frac": round(c["SQ_WAIT_ANY"] / wc, 3),
              "wait_inst_any_frac": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
              "active_inst_any_frac": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3),
              "lds_bank_conflict_frac": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 3)}
with open(f"profiles/{tag}_sq_summary.json", "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
