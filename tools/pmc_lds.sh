#!/bin/bash
# LDS pressure of the step's kernels: bank conflict cycles against the cycles the LDS pipe was active, per kernel (one rocprofv3 --pmc pass).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
lib=$1
OUT=/tmp/pmc_lds
rm -rf "$OUT"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d "$OUT" -- \
    python "$ROOT/tools/layout_sweep.py" --no-raster --steps 2 --lib "$ROOT/$lib" > "$ROOT/gpurun_out/lds.log" 2>&1
python - "$OUT" > "$ROOT/gpurun_out/lds.txt" <<'PY'
import csv, glob, collections, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("recon_", "mc_", "itx_add")):
            continue
        k = re.sub(r"\(anonymous namespace\)::|void |unsigned short|DevPlanes.*|\(.*", "", n)[:48]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    c = {a: sum(x) / len(x) for a, x in d[k].items()}
    w = max(c.get("SQ_WAVES", 1), 1)
    print("%-48s" % k, "  ".join("%s %.0f" % (a.replace("SQ_", ""), v / w) for a, v in sorted(c.items()) if a != "SQ_WAVES"), " (per wave)")
PY
cat "$ROOT/gpurun_out/lds.txt"; tail -2 "$ROOT/gpurun_out/lds.log"
