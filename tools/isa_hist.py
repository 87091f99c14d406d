#!/usr/bin/env python3
"""Static instruction histogram of one kernel in a device assembly file (hipcc -S --cuda-device-only).
usage: tools/isa_hist.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
c = collections.Counter()
for l in lines[start:end]:
    m = re.match(r"^\s+([a-z_0-9]+)\s", l)
    if m:
        c[m.group(1)] += 1
cls = collections.Counter()
for k, v in c.items():
    cls["VALU" if k.startswith("v_") else "SALU" if k.startswith("s_") else "LDS" if k.startswith("ds_") else
        "VMEM" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += v
print(sum(c.values()), dict(cls))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print("%5d %s" % (v, k))
