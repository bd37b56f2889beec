#!/bin/bash
# The driver's own command line is `bench.py --gpus 1 --steps 20 --warmup 5`: what do 20 timed steps read as a function of the
# number of warm-up steps queued directly ahead of them?  Alternating runs, one box.
for rep in 1 2 3 4; do
  for w in 0 5 20 100; do
    python3 bench.py --no-cpu --steps 20 --warmup $w 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('--warmup %-4s step %.3f us  value %.0f GB/s (%.3f of 8 TB/s)   kernel in chain %.2f us' % ('$w', j['ms_per_step']*1e3, j['value'], j['value']/8000, r['avg_kernel_us']))"
  done
done
