#!/bin/bash
# size of the preparation pool (DAV1D_HIP_PREP_THREADS) on the legs with 4 tile columns
for n in 4 6 8 12 4; do
  DAV1D_HIP_PREP_THREADS=$n python bench.py --steps 10 --warmup 2 --no-cpu --no-c1 --no-pmc --no-full > /tmp/p.json 2> /tmp/p.err
  python - $n <<'P'
import json,sys
d=json.load(open('bench_legs.json'))
out={}
for k in ('end_to_end_4_tile_columns','dav1d_task_loop','dav1d_task_loop_real_pass1'):
    v=d.get(k) or {}
    out[k]={x:v.get(x) for x in ('total_ms','list_ms','fps') if v.get(x) is not None}
fl=d.get('end_to_end_frames_in_flight') or {}
for k,v in fl.items():
    if isinstance(v,dict) and '4_tile' in k: out['in_flight_'+k]={x:v.get(x) for x in ('ms_per_frame','list_ms','host_cpu_ms_per_frame')}
print('prep_threads', sys.argv[1], json.dumps(out))
P
done
