#!/bin/bash
# the long sweeps: 2,048 AV1 streams through dav1d's own pass 1 (seeds 100 .. 2147) and 600 damaged streams, on the MI355X
mkdir -p gpurun_out/r06f
(DAV1D_STREAM_SEEDS=2048 timeout 2400 python -m pytest tests/test_stream.py -q -m gpu -k sweep) > gpurun_out/r06f/stream_sweep_2048_gpu.log 2>&1; tail -n 3 gpurun_out/r06f/stream_sweep_2048_gpu.log
(DAV1D_ERROR_SEEDS=600 timeout 900 python -m pytest tests/test_stream_errors.py -q -m gpu) > gpurun_out/r06f/error_sweep_600_gpu.log 2>&1; tail -n 3 gpurun_out/r06f/error_sweep_600_gpu.log
