#!/usr/bin/env python3
"""Prints gpurun_out/ko_*.json (tools/knockout.sh run) side by side."""
import glob, json, os, sys
names = sys.argv[1:] or sorted(os.path.basename(f)[3:-5] for f in glob.glob("gpurun_out/ko_*.json"))
for n in names:
    try:
        for line in open("gpurun_out/ko_%s.json" % n):
            d = json.loads(line)
            k = d.get("kernels_us", {})
            print("%-10s step %.4f  " % (n, d["ms_per_step"]) + "  ".join("%s %5.1f" % (a.replace("recon_", "r").replace("x" + a.split("x")[-1], "") if a.startswith("recon") else a, b) for a, b in k.items()) + "  sum %.1f" % sum(k.values()))
    except Exception as e:
        print(n, "failed:", e, open("gpurun_out/ko_%s.err" % n).read()[-300:])
