for b in 5 6 7 8; do for s in 11008,4096 13824,5120; do python tools/gemv_sweep.py --quick --shape $s,$b,128,1 --variants=-1 --splits 0,1; done; done
for b in 5 6 7 8; do python tools/gemv_sweep.py --quick --shape 4096,11008,$b,128,1 --variants=-1,109,108,112,114; done
for b in 3 5 7; do python tools/gemv_sweep.py --quick --shape 8192,8192,$b,128,1 --variants=-1,109,108; done
