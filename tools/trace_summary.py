#!/usr/bin/env python3
"""Per-dispatch table of a rocprofv3 --kernel-trace csv: start offset, duration, gap to the previous dispatch, short name.
usage: tools/trace_summary.py <kernel_trace.csv> [first_n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows[:n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print("%9.1f us  dur %7.1f  gap %6.1f  grid %7s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), name[:70]))
    prev_end = max(prev_end, e)
