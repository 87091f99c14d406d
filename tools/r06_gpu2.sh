#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_cdef.py tests/test_filmgrain.py tests/test_postchain.py tests/test_filter_lister.py tests/test_e2e.py tests/test_dsp_table.py -x -q -m gpu > gpurun_out/r06/gpu_tests2.log 2>&1; tail -2 gpurun_out/r06/gpu_tests2.log
for i in 1 2; do
python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-c1 --no-pmc --no-inflight > gpurun_out/r06/full3_$i.json 2> gpurun_out/r06/full3_$i.err
python - $i <<'P'
import json,sys
d=json.loads(open('gpurun_out/r06/full3_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d['legs']['full_table'])[:400])
P
done
