#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun).  Counters are collected in
# their own runs (one --pmc group per pass, kernel-trace only), as MI355X_MICROARCH.md prescribes.
# usage: tools/pmc_profile.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1   # per-kernel durations must not overlap: run the shapes back to back
ARGS="--steps 3 --warmup 1 --no-cpu --no-check --no-full --no-e2e --no-c1 --no-pmc $*"      # the recon step only: clean per-kernel averages
FULL_ARGS="--steps 3 --warmup 1 --no-cpu --no-check --no-e2e --no-c1 --no-inflight --no-pmc $*"         # + the full-table leg (post filters, intra waves)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full" -- python "$ROOT/bench.py" $FULL_ARGS > "$OUT/stats_full.log" 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python "$ROOT/bench.py" $ARGS > "$OUT/pmc$i.log" 2>&1
done
# the other stages of the full table (intra waves, deblock, CDEF, restoration, film grain): the same counters on the full-table leg
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_full$i" -- python "$ROOT/bench.py" $FULL_ARGS > "$OUT/pmc_full$i.log" 2>&1
done
# the step as it really runs (pipelined + paired launches on several streams): HBM bytes of everything it launches
unset DAV1D_HIP_SERIAL
STEP_ARGS="--steps 4 --warmup 0 --step-only $*"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/step_fetch" -- python "$ROOT/bench.py" $STEP_ARGS > "$OUT/step_fetch.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/step_write" -- python "$ROOT/bench.py" $STEP_ARGS > "$OUT/step_write.log" 2>&1
python "$ROOT/tools/pmc_summary.py" "$OUT" --step 4 > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
