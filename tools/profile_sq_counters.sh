#!/bin/bash
# Where the waves' cycles go: SQ counters of the dominant kernel of one bench.py workload (one rocprofv3 --pmc pass, no trace
# domains besides --kernel-trace).   bash tools/profile_sq_counters.sh OUT [bench.py args]
set -o pipefail
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$OUT/sq" -o bench -- python3 "$ROOT/bench.py" --no-cpu --steps 30 --warmup 5 --roofline-launches 200 "$@" > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
best = max(acc, key=lambda k: len(next(iter(acc[k].values()))))
with open(out + "/sq_counters.txt", "w") as o:
    o.write("# kernel: %s\n# launches: %d (averages per launch)\n" % (best, len(next(iter(acc[best].values())))))
    for c, v in sorted(acc[best].items()):
        o.write("%-28s %14.0f\n" % (c, sum(v) / len(v)))
print(open(out + "/sq_counters.txt").read())
PY
rm -rf "$OUT/sq"
