#!/bin/bash
# Shader clock while the step's kernels run: GRBM_GUI_ACTIVE (GPU cycles the kernel was active) over the kernel's duration from the trace.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
lib=$1
OUT=/tmp/pmc_clock
rm -rf "$OUT"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT" -- \
    python "$ROOT/tools/layout_sweep.py" --no-raster --steps 2 --lib "$ROOT/$lib" > "$ROOT/gpurun_out/clock.log" 2>&1
python - "$OUT" > "$ROOT/gpurun_out/clock.txt" <<'PY'
import csv, glob, collections, re, sys
dur = {}
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        d = dur.get(r["Dispatch_Id"])
        if not d or not any(k in d[1] for k in ("recon_", "mc_", "itx_add")):
            continue
        k = re.sub(r"\(anonymous namespace\)::|void |unsigned short|DevPlanes.*|\(.*", "", d[1])[:48]
        agg[k].append((float(r["Counter_Value"]), d[0]))
for k in sorted(agg):
    c = sum(a for a, b in agg[k]) / len(agg[k]); t = sum(b for a, b in agg[k]) / len(agg[k])
    print("%-48s cycles %9.0f  ns %8.0f  GHz %.2f" % (k, c, t, c / t))
PY
cat "$ROOT/gpurun_out/clock.txt"
