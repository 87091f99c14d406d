#!/bin/bash
# after the src_step fix, at HEAD: 2,048 streams with the lister's cell maps poisoned and 1,000 damaged streams
mkdir -p gpurun_out/r06j
(DAV1D_HIP_LISTER_POISON=1 DAV1D_STREAM_SEEDS=2048 timeout 2400 python -m pytest tests/test_stream.py -q -m gpu -k sweep) > gpurun_out/r06j/stream_sweep_2048_poisoned_maps_gpu.log 2>&1; tail -n 3 gpurun_out/r06j/stream_sweep_2048_poisoned_maps_gpu.log
(DAV1D_ERROR_SEEDS=1000 timeout 1200 python -m pytest tests/test_stream_errors.py -q -m gpu) > gpurun_out/r06j/error_sweep_1000_gpu.log 2>&1; tail -n 3 gpurun_out/r06j/error_sweep_1000_gpu.log
