#!/bin/bash
# The driver's command (--steps 20 --warmup 5): eager launches behind a blocker against a hipGraph replay of the same steps.
for rep in 1 2 3 4; do
  for a in "" "--graph"; do
    python3 bench.py --no-cpu --steps 20 --warmup 5 $a 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-8s step %.3f us  value %.0f GB/s (%.3f of 8 TB/s)   kernel in chain %.2f us' % ('$a' or 'eager', j['ms_per_step']*1e3, j['value'], j['value']/8000, r['avg_kernel_us']))"
  done
done
