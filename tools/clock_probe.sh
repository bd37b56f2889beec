#!/bin/bash
# Does a sustained back-to-back run lower the clocks?  Polls rocm-smi (sclk, mclk, power) every 20 ms while bench.py runs a long chain.
#   bash tools/clock_probe.sh OUT [bench.py args]
OUT=$1; shift
mkdir -p "$OUT"
python3 bench.py --no-cpu --steps 30000 --warmup 40 "$@" > "$OUT/bench_long.json.log" 2> "$OUT/bench_long.err.log" &
BP=$!
: > "$OUT/smi_poll.txt"
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> "$OUT/smi_poll.txt"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> "$OUT/smi_poll.txt"
  sleep 0.02
done
wait $BP
