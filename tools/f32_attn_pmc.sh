#!/bin/bash
# GPU box: issue / LDS counters of the fp32 mode's attention kernel (flash_attn_split_kernel) inside the fp32 forward of the headline shape,
# one rocprofv3 --pmc pass per counter group.   bash tools/f32_attn_pmc.sh <outdir>
set -e -o pipefail
OUT=${1:-gpurun_out/f32_attn_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- python3 tools/f32_bench_mx.py > $OUT/a.log 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/f32_bench_mx.py > $OUT/b.log 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 tools/f32_bench_mx.py > $OUT/c.log 2> $OUT/c.err
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for sub in "abc":
    cc = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    if not cc: print(sub, "no counters", open(os.path.join(out, sub + ".err")).read()[-400:]); continue
    for pat in ("flash_attn_split", "split_mxELb1", "Li4ELi0EDF16_Lb1"):
        acc = defaultdict(list); dur = []
        seen = set()
        for row in csv.DictReader(open(cc[0])):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"]); dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        if not dur: continue
        d = sum(dur) / len(dur)
        line = {k: sum(v) / len(v) for k, v in acc.items()}
        print(f"pass {sub} [{pat}]: launches {len(dur)} mean {d/1e3:.1f} us", " ".join(f"{k}={v:.4g}" for k, v in sorted(line.items())))
        if "GRBM_GUI_ACTIVE" in line:
            cyc = line["GRBM_GUI_ACTIVE"] / 8
            print(f"   clock = {cyc / d:.3f} GHz; per-SIMD cycles {cyc:.4g}")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in line: print(f"   MFMA busy = {line['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} of SIMD cycles")
            if "SQ_WAVE_CYCLES" in line:
                wc = line["SQ_WAVE_CYCLES"]
                print("   of wave cycles:", " ".join(f"{k}={line[k] / wc:.3f}" for k in line if k.startswith("SQ_WAIT") or k.startswith("SQ_ACTIVE")))
PY
find $OUT -name "*kernel_trace.csv" -delete
