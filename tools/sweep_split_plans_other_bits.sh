for b in 5 6 8; do for s in 11008,4096 13824,5120; do
  python tools/gemv_sweep.py --quick --shape $s,$b,128,1 --variants=-1 --splits 0,1
  python tools/gemv_sweep.py --quick --shape $s,$b,128,1 --variants=202,205 --splits 2,3,4,5,6,8,11
done; done
for b in 2 3; do for s in 11008,4096 8192,3584; do
  python tools/gemv_sweep.py --quick --shape $s,$b,128,1 --variants=-1 --splits 0,1
  python tools/gemv_sweep.py --quick --shape $s,$b,128,1 --variants=202,205 --splits 2,3,4,6
done; done
