set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r4f; mkdir -p $O
for cfg in "8 0" "12 0" "12 8" "12 4"; do set -- $cfg; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$1_$2 -- python3 tools/gemm_traffic.py --variant $1 --raster $2 > $O/f_$1_$2.log 2>&1; done
python3 tools/gemm_traffic_summary.py $O/f_8_0 $O/f_12_0 $O/f_12_8 $O/f_12_4 > $O/summary.txt
find $O -name "*kernel_trace.csv" -delete; cat $O/summary.txt
