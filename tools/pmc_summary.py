#!/usr/bin/env python3
"""Condense the rocprofv3 csv outputs of tools/pmc_profile.sh into one table per kernel:
average duration (kernel-trace stats) and per-launch counter averages."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"(mc_kernel|itx_add_kernel|comp_kernel|\w+_kernel)<([^>]*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace("unsigned short", "u16").replace("unsigned char", "u8").replace(" ", ""))
    return name[:40]


stats = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
for f in glob.glob(os.path.join(out, "stats_full", "**", "*kernel_stats.csv"), recursive=True):      # stages outside the recon step
    for r in csv.DictReader(open(f)):
        stats.setdefault(short(r["Name"]), (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(stats, key=lambda k: -stats[k][2])
print("%-44s %6s %10s %6s" % ("kernel", "calls", "avg_us", "pct"))
for k in names:
    print("%-44s %6d %10.1f %6.2f" % (k, *stats[k]))
print()
for k in names:
    if k not in cnt:
        continue
    print(k)
    for c in sorted(cnt[k]):
        v = cnt[k][c]
        print("    %-24s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))

# per-kernel HBM traffic per launch.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950
# FETCH_SIZE tallies 128-B read requests as 64 B for wide coalesced streams (MI355X_MICROARCH.md,
# "HBM"): both the raw and the doubled figure are kept, ratios between kernels are unaffected.
import json
traffic = {}
for k in names:
    c = cnt.get(k, {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
        traffic[k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "avg_us": stats[k][1]}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)

# the step as it really runs: bytes of every kernel the `--step-only` passes launched (step_fetch / step_write), per step
if "--step" in sys.argv:
    n_steps = int(sys.argv[sys.argv.index("--step") + 1])
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    per_kernel = defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
    for d, cname in (("step_fetch", "FETCH_SIZE"), ("step_write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cname:
                    continue
                k = short(r["Kernel_Name"])
                if not re.match(r"(mc_kernel|mc_twin_kernel|itx_add_kernel|itx_add_wide_kernel|recon_fused_kernel|comp_kernel|mc_all_kernel|itx_multi_kernel)", k):
                    continue        # fills / copies of the harness are not the step
                v = float(r["Counter_Value"]) * 1024
                tot[cname] += v
                per_kernel[k][cname] += v
                if cname == "FETCH_SIZE":
                    per_kernel[k]["launches"] += 1
    if tot["FETCH_SIZE"] and tot["WRITE_SIZE"]:
        step = {"steps": n_steps, "bytes_per_step": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / n_steps,
                "fetch_bytes_x2_per_step": 2 * tot["FETCH_SIZE"] / n_steps, "write_bytes_per_step": tot["WRITE_SIZE"] / n_steps,
                "kernels": {k: {"launches_per_step": v["launches"] / n_steps,
                                "bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / max(v["launches"], 1)}
                            for k, v in sorted(per_kernel.items())}}
        json.dump(step, open(os.path.join(out, "step_traffic.json"), "w"), indent=1)
        print()
        print("step traffic: %.1f MB per step (fetch x2 %.1f MB + write %.1f MB)" % (step["bytes_per_step"] / 1e6,
              step["fetch_bytes_x2_per_step"] / 1e6, step["write_bytes_per_step"] / 1e6))
