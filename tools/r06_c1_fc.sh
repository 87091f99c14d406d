#!/bin/bash
# 4K 8-bit (BASELINE configs[1]) step over 1 .. 6 frame contexts
for n in 1 2 3 4 6; do
  for mix in c1; do
    python bench.py --width 3840 --height 2160 --bpc 8 --mix $mix --step-only --frame-contexts $n --steps 60 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fc $n mix $mix', d['ms_per_step'], d['value'])"
  done
done
