#!/bin/bash
# quick: headline only; prints value, ms, kernel table
python bench.py --no-cpu --no-full --no-e2e "$@" > gpurun_out/qb.json 2> gpurun_out/qb.err || tail -5 gpurun_out/qb.err
python - <<PY
import json
d = json.loads(open("gpurun_out/qb.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "dom", d["roofline"]["kernel"], d["roofline"]["frac"], "path", d["roofline"]["path"]["frac"], d["config"]["parity"][:20])
print(d["roofline"]["kernels_ms"])
PY
