#!/bin/bash
# Knock-out variants of the reconstruction kernels (timing only: the pixels of a knocked-out build are wrong).  Each variant removes ONE
# thing a wave does — the window gathers, the tap table loads, the coefficient loads, the stores of the finished blocks, the second
# prediction of compound blocks, the filter arithmetic, the transform arithmetic — and tools/layout_sweep.py --kernels times the step's
# launches with it: the difference to the plain build is the most a rewrite of that part can give.
#   tools/knockout.sh build "NAME=-DDV_KO_A -DDV_KO_B" ...   |   tools/knockout.sh run "<layout_sweep sets>" NAME ...
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
    shift
    for spec in "$@"; do python tools/build_variant.py "ko_${spec%%=*}" "-DDV_LEAN ${spec#*=}" recon.hip mc.hip itx.hip capi.hip & done
    wait
else
    sets="$2"
    shift; shift
    mkdir -p gpurun_out
    for v in "$@"; do
        lib=dav1d_amd/build/variants/ko_$v.so
        python tools/layout_sweep.py --kernels --no-raster --lib $lib --sets $sets > gpurun_out/ko_$v.json 2>gpurun_out/ko_$v.err || true
    done
fi
