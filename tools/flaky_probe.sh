#!/bin/bash
# Tests N times under load (16 workers on one GPU): failures by name.  tools/flaky_probe.sh [dir] [N] [pytest target ...]
cd "${1:-.}"; n=${2:-5}; shift; shift
for i in $(seq $n); do python -m pytest ${@:-tests} -q -m gpu -n 16 --tb=line 2>&1 | grep -E "^(FAILED|[0-9]+ (passed|failed))" | cut -c1-170 | head -6; done
