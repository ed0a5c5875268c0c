#!/bin/bash
for sr in 8 16 32; do for g in 4096 16384; do
  FDTD2D_PML_SHORT=$sr python bench.py --grid $g --boundary pml --steps 96 --warmup 16 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('short', $sr, d['config']['grid'], d['value'])"
done; done
