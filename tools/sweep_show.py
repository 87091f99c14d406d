#!/usr/bin/env python3
"""Prints a tools/layout_sweep.py output file compactly."""
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try:
            d = json.loads(l)
        except ValueError:
            continue
        k = d.get("kernels_us") or {}
        eq = d.get("equals_raster", d.get("equals_first"))
        print("%-22s %-20s step %.4f eq %-5s %s " % (f.split("/")[-1][:22], d.get("opts"), d["ms_per_step"], eq, d.get("digest", "")) + " ".join("%s %.1f" % (a.replace("recon_", "r").split("x")[0] if a.startswith("recon") else a, b) for a, b in k.items()))
