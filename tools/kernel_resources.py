#!/usr/bin/env python3
"""Register / LDS / scratch budget and occupancy of every kernel of one HIP source, as the compiler reports it
(-Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.py recon.hip [mc.hip ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


for src in sys.argv[1:]:
    path = os.path.join(ROOT, "dav1d_amd", "csrc", src)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            continue
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
        if "lds" in cur and "name" in cur:
            n = demangle(cur["name"])
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            n = n.split("(")[0]
            print("%-12s vgpr %3d agpr %3d scratch %4d lds %6d occ %2d  %s" % (src, cur.get("vgpr", -1), cur.get("agpr", -1), cur.get("scratch", -1),
                                                                               cur["lds"], cur.get("occ", -1), n[:100]))
            cur = {}
