set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for mx in 0 1; do
O=gpurun_out/r4l_mx$mx; mkdir -p $O
RZ_MX=$mx rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/f32_bench_mx.py > $O/out.txt 2> $O/err.txt
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; find $O/trace -name "*kernel_trace.csv" -delete
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x {int(r["Calls"]):5d}  {float(r["Percentage"]):5.1f}%  {r["Name"][:120]}')
PY
done
