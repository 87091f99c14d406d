#!/bin/bash
# one or more rocprofv3 counter passes over a short bench run, every pass under its own timeout
# usage: tools/pmc_quick.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]   (env is passed through to bench.py)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --no-check > "$OUT/pmc$i.log" 2>&1
done
python - <<PY
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        k = "mc_all" if "mc_all" in n else n.replace("void (anonymous namespace)::", "")[:34]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("%-36s" % k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
