#!/bin/bash
# full-table stages only; prints stages_ms
python bench.py --no-cpu --no-e2e "$@" > gpurun_out/sb.json 2> gpurun_out/sb.err || tail -5 gpurun_out/sb.err
python - <<PY
import json
d = json.loads(open("gpurun_out/sb.json").read().strip().splitlines()[-1])
ft = d.get("full_table") or d["config"].get("full_table")
print("full", ft["ms_per_frame"], "frac", ft["frac"], ft["stages_ms"], ft["parity"][:30])
PY
