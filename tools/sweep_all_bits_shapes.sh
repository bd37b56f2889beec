# every bit-width on a set of decoder-layer shapes, automatic launch: a table to spot launch-rule outliers
for s in 4096,4096 4096,11008 4096,14336 11008,4096 14336,4096 8192,8192 8192,28672 28672,8192 5120,13824 5120,5120 8192,1024; do
  for b in 2 3 4 5 6 7 8; do python tools/gemv_sweep.py --quick --shape $s,$b,128,1 --variants=-1; done
done
