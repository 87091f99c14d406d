"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: how much of a bench step the kernels cover,
how much of that is concurrent, and where the gaps are.  usage: timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(mc_kernel|itx_add_kernel|comp_kernel|recon_fused_kernel)<([^>]*)>", name)
    return (m.group(1).replace("_kernel", "") + "<" + m.group(2).replace("unsigned short", "u16").replace(" ", "") + ">") if m else name[:40]


def main():
    path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
    # a step starts at every first mc kernel after an itx kernel
    steps, cur, seen_itx = [], [], False
    for e in ev:
        is_mc = e[2].startswith("mc")
        if is_mc and seen_itx:
            steps.append(cur); cur = []; seen_itx = False
        if e[2].startswith("itx"):
            seen_itx = True
        cur.append(e)
    steps.append(cur)
    for si, st in enumerate(steps[-3:]):
        st = [e for e in st if e[2].startswith(("mc", "itx", "comp", "recon"))]
        if not st:
            continue
        t0, t1 = min(e[0] for e in st), max(e[1] for e in st)
        busy = 0
        edge = sorted([(e[0], 1) for e in st] + [(e[1], -1) for e in st])
        depth, last, conc = 0, t0, 0
        for t, d in edge:
            if depth > 0:
                busy += t - last
            if depth > 1:
                conc += t - last
            depth += d
            last = t
        print("step %d: span %.1f us, covered %.1f us, >=2 kernels %.1f us, sum of kernels %.1f us, n=%d" %
              (si, (t1 - t0) / 1e3, busy / 1e3, conc / 1e3, sum(e[1] - e[0] for e in st) / 1e3, len(st)))
        if si == len(steps[-3:]) - 1:
            for e in st:
                print("   %8.1f %8.1f  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))


if __name__ == "__main__":
    main()
