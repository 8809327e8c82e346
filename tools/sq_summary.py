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
    if not any(s in k for s in ("chain", "conv_", "stem", "trp", "bilinear", "crop")):
        continue
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    out[k] = {"launches": cnt[k],
              "wait_any_frac": round(c["SQ_WAIT_ANY"] / wc, 3),
              "wait_inst_any_frac": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
              "active_inst_any_frac": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3),
              "lds_bank_conflict_frac": round(c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), 3)}
with open(f"profiles/{tag}_sq_summary.json", "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
