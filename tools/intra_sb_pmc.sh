#!/bin/bash
# PMC passes over the key-frame leg of tools/intra_sb_probe.py (through gpurun): where the superblock kernel's wave cycles go.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-sb_pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/intra_sb_probe.py --modes 2 --lds 0 --flow 1 --no-pass --no-check"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SMEM" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc*")):
    if not d.rsplit("/",1)[1][3:].isdigit(): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "intra_sb" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            agg[k]["_n"] += 0
    for k, v in agg.items():
        print(d.rsplit("/",1)[1], k, {a: round(b) for a, b in v.items() if a != "_n"})
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "intra" in r["Name"]: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
