# 16-wave builds that exceed 128 VGPRs (spills) against 8 waves over two sweeps, tuning library: 121 = up to 16 waves, 120 = at most 8
for m in 1 4; do
  for b in 3; do for s in 8192,8192 8192,14336; do python tools/gemv_sweep.py --quick --shape $s,$b,128,$m --variants=121,120; done; done
  for b in 5 6 7; do for s in 4096,14336 8192,8192; do python tools/gemv_sweep.py --quick --shape $s,$b,128,$m --variants=121,120; done; done
  python tools/gemv_sweep.py --quick --shape 4096,11008,7,128,$m --variants=121,120
done
for b in 4 8; do for s in 8192,8192 8192,14336 4096,14336; do python tools/gemv_sweep.py --quick --shape $s,$b,128,4 --variants=121,120; done; done
