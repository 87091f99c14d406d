#!/bin/bash
# rocprofv3 counter passes over tools/twin_probe.py (raster and tiled references in one process: the kernel names differ by their
# last template argument).  usage: tools/pmc_probe.sh <tag> [probe args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python "$ROOT/tools/twin_probe.py" --steps 3 "$@" > "$OUT/pmc$i.log" 2>&1
done
python - <<PY
import csv, glob, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "recon_fused" not in n and "mc_kernel" not in n and "itx_add" not in n and "retile" not in n:
            continue
        k = re.sub(r"\(anonymous namespace\)::|void |unsigned short|DevPlanes.*", "", n)[:60]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print("%-50s" % k, {c: round(sum(x) / len(x)) for c, x in sorted(d[k].items())})
PY
