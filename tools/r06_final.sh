#!/bin/bash
# the round's last call: GPU suite, the default bench line, the driver's own command (20 steps behind 5 warm-up steps), rocprofv3 statistics + counters
D=${1:-r06h}
mkdir -p gpurun_out/$D
(time python -m pytest tests -x -q -m gpu) > gpurun_out/$D/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$D/gpu_tests.log
python bench.py > gpurun_out/$D/bench.json 2> gpurun_out/$D/bench.err; cp bench_legs.json gpurun_out/$D/bench_legs.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-c1 --no-inflight > gpurun_out/$D/bench_driver_command_20_steps.json 2> gpurun_out/$D/bench20.err
timeout 1500 bash tools/pmc_profile.sh ${D}_pmc > gpurun_out/$D/pmc.log 2>&1
tail -n 3 gpurun_out/$D/gpu_tests.log; cut -c1-600 gpurun_out/$D/bench.json; cut -c1-400 gpurun_out/$D/bench_driver_command_20_steps.json
