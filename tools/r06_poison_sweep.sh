#!/bin/bash
# after the src_step fix: the streams that failed, then 640 streams with the lister's cell maps poisoned (a step drawn from a cell outside the share its tile cleared = -ERANGE)
mkdir -p gpurun_out/r06g
(python -m pytest tests/test_stream.py tests/test_lister.py -q -m gpu -k "wavefront_steps_come or steps_of_intra_block") > gpurun_out/r06g/new_tests_gpu.log 2>&1; tail -n 2 gpurun_out/r06g/new_tests_gpu.log
(DAV1D_HIP_LISTER_POISON=1 DAV1D_STREAM_SEEDS=640 python -m pytest tests/test_stream.py -q -m gpu -k sweep) > gpurun_out/r06g/stream_sweep_640_poisoned_maps_gpu.log 2>&1; tail -n 2 gpurun_out/r06g/stream_sweep_640_poisoned_maps_gpu.log
