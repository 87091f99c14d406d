#!/bin/bash
# A/B of variant libraries in ONE gpurun call (boxes differ by 10 %): tools/ab.sh [-r reps] name ...  -> gpurun_out/ab_<name>.json
# (each name = dav1d_amd/build/variants/ko_<name>.so of tools/knockout.sh build; the first one is also the pixel reference of the others)
cd "$(dirname "$0")/.."
reps=1
if [ "$1" = -r ]; then reps=$2; shift; shift; fi
mkdir -p gpurun_out
for r in $(seq $reps); do
for v in "$@"; do
    python tools/layout_sweep.py --kernels --no-raster --lib dav1d_amd/build/variants/ko_$v.so ${SWEEP_SETS:+--sets $SWEEP_SETS} --digest >> gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err || tail -3 gpurun_out/ab_$v.err
done
done
