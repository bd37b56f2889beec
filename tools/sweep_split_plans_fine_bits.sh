# The fine split-K plan sweep of tools/sweep_split_plans_fine.sh for another bit-width (a data set for the planner):
#   bash tools/sweep_split_plans_fine_bits.sh 5
b=${1:-5}
for shape in 8192,3584 11008,4096 14336,4096 28672,4096 28672,2048 16384,1024 8192,2048 8192,4096 13824,5120 8192,1024 13824,2560 11008,2048 8192,5120 28672,1024; do
  python tools/gemv_sweep.py --quick --shape $shape,$b,128,1 --variants=-1 --splits 0,1
  python tools/gemv_sweep.py --quick --shape $shape,$b,128,1 --variants=202,205 --splits 2,3,4,5,6,7,8,9,10,11,12,13
done
