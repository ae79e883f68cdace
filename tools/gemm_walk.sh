#!/bin/bash
# GPU box (RZ_EXPERIMENTS=1 library): VERDICT r5 item 5 — gemm8's tile walk with groups of G row tiles (option gemm_raster = 100 + G; shipped: 4),
# judged by bytes (FETCH_SIZE per instantiation) AND inside the step (tools/step_ab.py, bit-identity checked).   bash tools/gemm_walk.sh
set -e -o pipefail
export RZ_EXPERIMENTS=1
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06/walk; mkdir -p $O
python3 tools/step_ab.py "gemm_raster=0" "gemm_raster=101" "gemm_raster=102" "gemm_raster=108" "gemm_raster=116" --rounds 5 --steps 6 --check --json $O/step_ab.json > $O/step_ab.txt 2>&1
tail -12 $O/step_ab.txt
for r in 0 101 102 108; do rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$r -- python3 tools/gemm_traffic.py --variant 8 --raster $r > $O/f_$r.log 2>&1; done
python3 tools/gemm_traffic_summary.py $O/f_0 $O/f_101 $O/f_102 $O/f_108 > $O/fetch_summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/fetch_summary.txt
