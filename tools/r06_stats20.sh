#!/bin/bash
# rocprofv3 --kernel-trace --stats of the recon step's shapes back to back: ONE frame context and no side streams (DAV1D_HIP_SERIAL=1), so that no two launches overlap and a
# kernel's average is its duration alone (with two frame contexts the launches of two frames run side by side and each lasts longer: kernel_stats_two_frames_in_flight.csv)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06h_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DAV1D_HIP_SERIAL=1
for rep in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fc1_$rep" -- python "$ROOT/bench.py" --steps 20 --warmup 5 --frame-contexts 1 --no-cpu --no-check --no-full --no-e2e --no-c1 --no-pmc > "$OUT/stats_fc1_$rep.log" 2>&1
cp $OUT/stats_fc1_$rep/*/*_kernel_stats.csv $OUT/kernel_stats_one_frame_at_a_time_$rep.csv
tail -1 $OUT/stats_fc1_$rep.log | cut -c1-200
done
