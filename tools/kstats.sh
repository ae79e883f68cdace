#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of a short default-workload bench run with extra bench flags; prints the top kernels.
#   bash tools/kstats.sh <tag> [bench flags...]     -> gpurun_out/<tag>/kernel_stats.csv
set -e -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench.json 2> $OUT/trace.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -size +20M -delete || true
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n = r["Name"]
    n = n.replace("void rz::", "").replace("rz::", "")
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {int(r["Calls"]):5d}  {float(r["Percentage"]):5.1f}%  {n[:110]}')
PY
