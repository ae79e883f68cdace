#!/bin/bash
# GPU box: per-launch durations (rocprofv3 kernel trace) of the split GEMM forms over K  -> gpurun_out/r4n/
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out/r4n
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4n/trace -- python3 tools/mx_gemm_time.py > gpurun_out/r4n/out.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4n/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
i = 0
for K in (256, 768, 1536, 3072):
    for form in (0, 1):
        chunk = seq[i:i + 7]; i += 7
        print(f"K={K} form {form}: median {sorted(c[1] for c in chunk)[3]:.1f} us   {chunk[0][0][:80]}")
PY
