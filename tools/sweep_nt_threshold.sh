# nontemporal (122) against plain (123) weight loads on the default path, by weight bytes and bit-width (tuning library)
for spec in 4096,4096,8 4096,11008,8 4096,14336,8 8192,8192,8 5120,13824,8 4096,11008,6 4096,14336,6 8192,8192,6 4096,11008,5 8192,8192,5 4096,11008,7 4096,11008,4 4096,14336,4 8192,8192,4 8192,14336,4 5120,13824,4 4096,22016,4 4096,11008,3 8192,14336,3 4096,11008,2 8192,14336,2; do
  IFS=, read k n b <<< "$spec"
  python tools/gemv_sweep.py --quick --shape $k,$n,$b,128,1 --variants=123,122
done
