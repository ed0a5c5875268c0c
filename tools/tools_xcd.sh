#!/bin/bash
# A/B of the XCD-aware strip order (FDTD2D_XCD_MAP), separate processes on one box, alternating
for rep in 1 2; do
for m in 0 1; do
  for cfg in "--grid 4096" "--grid 8192" "--grid 16384 --steps 96 --warmup 32" "--grid 4096 --cols 32768 --steps 96 --warmup 32" "--grid 8192 --materials ring --steps 96 --warmup 32"; do
    FDTD2D_XCD_MAP=$m python bench.py --no-cpu-baseline $cfg | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('xcd_map=$m', d['config']['grid'], d['config']['materials'], 'value', d['value'], 'launch_ms', r['avg_launch_ms'], r['launch_shape'])"
  done
done
done
