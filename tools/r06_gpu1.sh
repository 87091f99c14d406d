mkdir -p gpurun_out/r06c
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r06c/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06c/gpu_tests.log
python bench.py > gpurun_out/r06c/bench.json 2> gpurun_out/r06c/bench.err; cp bench_legs.json gpurun_out/r06c/bench_legs.json 2>/dev/null
timeout 1500 bash tools/pmc_profile.sh r06c_pmc > gpurun_out/r06c/pmc.log 2>&1
tail -3 gpurun_out/r06c/gpu_tests.log; cut -c1-600 gpurun_out/r06c/bench.json
