#!/bin/bash
# chunk preparation on the library's threads (prep_async = 1, frames of <= 8 tiles) against on the listing thread (0): the legs with 4 tile columns and dav1d's loop
mkdir -p gpurun_out/r06
python -m pytest tests/test_e2e.py tests/test_lister.py tests/test_hooked.py tests/test_stream.py tests/test_stream_errors.py tests/test_refmvs.py -x -q -m gpu > gpurun_out/r06/gpu_tests3.log 2>&1; tail -2 gpurun_out/r06/gpu_tests3.log
for m in 0 1 0 1; do
  DAV1D_HIP_PREP_ASYNC=$m python bench.py --steps 10 --warmup 2 --no-cpu --no-c1 --no-pmc > gpurun_out/r06/prep_$m.json 2> gpurun_out/r06/prep_$m.err
  python - $m <<'P'
import json,sys
d=json.load(open('bench_legs.json'))
keys=('end_to_end','end_to_end_4_tile_columns','end_to_end_full_table_4_tile_columns','dav1d_task_loop','dav1d_task_loop_real_pass1')
out={}
for k in keys:
    v=d.get(k) or {}
    out[k]={x:v.get(x) for x in ('total_ms','list_ms','fps','peer_fps') if v.get(x) is not None}
fl=d.get('end_to_end_frames_in_flight') or {}
for k,v in fl.items():
    if isinstance(v,dict) and '4_tile' in k: out['in_flight_'+k]={x:v.get(x) for x in ('ms_per_frame','list_ms','host_cpu_ms_per_frame')}
print('prep_async', sys.argv[1], json.dumps(out))
P
done
