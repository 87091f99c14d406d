#!/bin/bash
# Instructions per wave, by kind, of the step's kernels (one rocprofv3 --pmc pass per variant library over tools/layout_sweep.py): with the
# knock-out variants of tools/knockout.sh the differences say how many instructions each part of a wave's program costs.
#   tools/pmc_insts.sh <variant.so> ...      -> gpurun_out/insts_<variant>.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
for lib in "$@"; do
    tag=$(basename "$lib" .so)
    OUT=/tmp/pmc_$tag
    rm -rf "$OUT"
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT" -- \
        python "$ROOT/tools/layout_sweep.py" --no-raster --steps 2 --lib "$ROOT/$lib" ${SWEEP_SETS:+--sets $SWEEP_SETS} > "$ROOT/gpurun_out/insts_$tag.log" 2>&1
    python - "$OUT" > "$ROOT/gpurun_out/insts_$tag.txt" <<'PY'
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
    print("%-48s waves %6d  per wave: VALU %6.0f SALU %6.0f LDS %5.0f SMEM %4.0f VMEM %4.0f  cycles/wave %7.0f  valu_active/wave %6.0f" % (
        k, w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w, c.get("SQ_INSTS_SMEM", 0) / w,
        c.get("SQ_INSTS_VMEM", 0) / w, 4 * c.get("SQ_WAVE_CYCLES", 0) / w, 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / w))
PY
done
