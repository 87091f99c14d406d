#!/bin/bash
# A/B of library builds on the recon step (GPU box, through gpurun): tools/ab_step.sh <variant.so | ""> ...
# every argument is a library under dav1d_amd/build/variants/ ("" = the tree's own build); the step's parity gate stays on
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
cp dav1d_amd/libdav1d_hip.so /tmp/libdav1d_hip_tree.so
for v in "$@"; do
    if [ -n "$v" ]; then cp "dav1d_amd/build/variants/$v" dav1d_amd/libdav1d_hip.so; else cp /tmp/libdav1d_hip_tree.so dav1d_amd/libdav1d_hip.so; fi
    timeout 200 python bench.py --no-cpu --no-e2e --no-c1 --no-full $AB_ARGS > /tmp/ab.json 2> /tmp/ab.err || { echo "[$v] FAILED: $(tail -3 /tmp/ab.err)"; continue; }
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
r = d["roofline"]
print("[%s] ms_per_step %.4f  path_frac %.4f  kernels_ms %s  parity: %s" % (sys.argv[1] or "tree", d["ms_per_step"], r.get("path_frac", 0), r.get("kernels_ms"), str(d.get("parity", d.get("config", {}).get("parity", "")))[:60]))
PY
done
cp /tmp/libdav1d_hip_tree.so dav1d_amd/libdav1d_hip.so
