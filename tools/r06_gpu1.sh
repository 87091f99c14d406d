mkdir -p gpurun_out/r06d
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r06d/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06d/gpu_tests.log
python bench.py > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err; cp bench_legs.json gpurun_out/r06d/bench_legs.json 2>/dev/null
timeout 1500 bash tools/pmc_profile.sh r06d_pmc > gpurun_out/r06d/pmc.log 2>&1
tail -3 gpurun_out/r06d/gpu_tests.log; cut -c1-600 gpurun_out/r06d/bench.json
