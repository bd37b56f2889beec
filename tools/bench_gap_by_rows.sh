for rep in 1 2; do for m in 1 2 4 8 16; do python3 bench.py --no-cpu --shape 4096,11008,4,128,$m 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('M=%-3s step %.2f us  kernel in chain %.2f  isolated %.2f   gap %.2f' % ('$m', j['ms_per_step']*1e3, r['avg_kernel_us'], r['isolated_launch_avg_kernel_us'], j['ms_per_step']*1e3-r['avg_kernel_us']))"; done; done
