#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/gpu_pmc.sh (FETCH_SIZE and WRITE_SIZE, each in its own run)
into profiles/<tag>_pmc_summary.json: MB per launch and kernel family.  FETCH_SIZE is doubled, as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (it tallies 128-B requests at 64 B);
both counters are in KiB."""
import collections
import csv
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
FAM = (("fuse_module_kernel", "fuse_module_kernel"), ("fuse_head4_kernel", "fuse_head4_kernel"), ("conv_pair_next", "conv_pair_next(+cat)"),
       ("conv_pair_kernel", "conv_pair_kernel"), ("conv_dual_kernel", "conv_dual_kernel"), ("stem_persist_kernel", "stem_persist_kernel"), ("stem_kernel", "stem_kernel"),
       ("chain_group_wide_kernel", "chain_group_wide_kernel"), ("chain_group_kernel", "chain_group_kernel"), ("chain_kernel", "chain_kernel"), ("conv_group_kernel", "conv_group_kernel"),
       ("conv_igemm_kernel", "conv_igemm_kernel"), ("nchw_to_s2d16_kernel", "nchw_to_s2d16_kernel"), ("trp_", "trp kernels"))


def fam(name):
    for key, label in FAM:
        if key in name:
            return label
    return "other"


def read(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                k = fam(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


ft, fc = read("gpurun_out/pmc_fetch/pmc_counter_collection.csv", "FETCH_SIZE")
wt, wc = read("gpurun_out/pmc_write/pmc_counter_collection.csv", "WRITE_SIZE")
out = {}
for k in ft:
    out[k] = {"launches": fc[k], "fetch_mb_per_launch": round(ft[k] * 1024 * 2 / fc[k] / 1e6, 2),
              "write_mb_per_launch": round(wt[k] * 1024 / max(wc[k], 1) / 1e6, 2)}
with open(f"profiles/{tag}_pmc_summary.json", "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
