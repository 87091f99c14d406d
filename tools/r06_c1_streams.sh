#!/bin/bash
# 4K 8-bit (BASELINE configs[1]) step: paired launches on 0 / 1 / 2 side streams x frame contexts (is the leg bound by enqueueing?), then 1080p
mkdir -p gpurun_out/r06e
for rep in 1 2; do
for ps in 2 1 0; do
  for n in 1 2 3; do
    DAV1D_HIP_RECON_PAIR_STREAMS=$ps python bench.py --width 3840 --height 2160 --bpc 8 --mix c1 --step-only --frame-contexts $n --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K pair_streams $ps fc $n', d['ms_per_step'], d['value'])"
  done
done
done 2>&1 | tee gpurun_out/r06e/c1_streams.txt
for ps in 2 0; do
  for n in 1 2; do
    DAV1D_HIP_RECON_PAIR_STREAMS=$ps python bench.py --width 1920 --height 1080 --bpc 8 --mix c1 --step-only --frame-contexts $n --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p pair_streams $ps fc $n', d['ms_per_step'], d['value'])"
  done
done 2>&1 | tee -a gpurun_out/r06e/c1_streams.txt
# the longer sweeps at HEAD
(DAV1D_STREAM_SEEDS=640 python -m pytest tests/test_stream.py -q -m gpu -k sweep) > gpurun_out/r06e/stream_sweep_640_gpu.log 2>&1; tail -n 2 gpurun_out/r06e/stream_sweep_640_gpu.log
(DAV1D_ERROR_SEEDS=200 python -m pytest tests/test_stream_errors.py -q -m gpu) > gpurun_out/r06e/error_sweep_200_gpu.log 2>&1; tail -n 2 gpurun_out/r06e/error_sweep_200_gpu.log
