python tools/gemv_sweep.py --quick --shape 4096,4096,4,128,1 --variants=-1,190,191,192
python tools/gemv_sweep.py --quick --shape 8192,3584,4,128,1 --variants=-1,190,191
python tools/gemv_sweep.py --quick --shape 8192,3584,4,128,1 --variants=206 --splits 2,3,4
python tools/gemv_sweep.py --quick --shape 4096,4096,4,128,1 --variants=206 --splits 2,3
python tools/gemv_sweep.py --quick --shape 11008,4096,4,128,1 --variants=-1,190
python tools/gemv_sweep.py --quick --shape 11008,4096,4,128,1 --variants=206 --splits 2,3,4
python tools/gemv_sweep.py --quick --shape 8192,8192,4,128,1 --variants=-1,190,191
python tools/gemv_sweep.py --quick --shape 4096,11008,4,128,1 --variants=-1,190
