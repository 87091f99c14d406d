#!/bin/bash
# Start / duration of every kernel of the last steps of tools/layout_sweep.py as they really run (streams side by side): rocprofv3 --kernel-trace.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/steptrace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/steptrace -- python "$ROOT/tools/layout_sweep.py" --no-raster --steps 6 ${1:+--lib $ROOT/$1} > "$ROOT/gpurun_out/steptrace.log" 2>&1
python - <<'PY' > "$ROOT/gpurun_out/steptrace.txt"
import csv, glob, re
p = glob.glob("/tmp/steptrace/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(p)):
    n = r["Kernel_Name"]
    if not any(k in n for k in ("recon_", "mc_", "itx_add")):
        continue
    k = re.sub(r"\(anonymous namespace\)::|void |unsigned short|DevPlanes.*|\(.*", "", n)[:44]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "")))
ev.sort()
# steps: 7 kernels each
for s in range(len(ev) // 7 - 2, len(ev) // 7):
    st = ev[s * 7:(s + 1) * 7]
    t0 = min(e[0] for e in st)
    print("step", s, "span %.1f us" % ((max(e[1] for e in st) - t0) / 1e3))
    for e in st:
        print("   start %7.1f  dur %6.1f  end %7.1f  q %s  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[1] - t0) / 1e3, e[3], e[2]))
PY
cat "$ROOT/gpurun_out/steptrace.txt"
