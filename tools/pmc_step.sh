#!/bin/bash
# counter passes over the recon step alone (launches serialised so that per-kernel counters do not mix)
# usage: tools/pmc_step.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
i=0
for grp in "$@"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --no-check --no-full --no-e2e --no-c1 > "$OUT/pmc$i.log" 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $OUT/pmc$i.log)"
done
python - <<PY
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        k = n.replace("void (anonymous namespace)::", "")[:40]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print("%-42s" % k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
