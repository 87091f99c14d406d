#!/bin/bash
# rocprofv3 kernel statistics of tools/twin_probe.py (serial launches): usage tools/stats_probe.sh <tag> [probe args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$ROOT/tools/twin_probe.py" --steps 5 --no-kernels "$@" > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv, re
for r in csv.DictReader(open("$f")):
    n = re.sub(r"\(anonymous namespace\)::|void |unsigned short|DevPlanes.*", "", r["Name"])[:70]
    if "recon" in n or "mc_" in n or "itx" in n or "retile" in n:
        print("%-72s calls %4s avg %8.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
