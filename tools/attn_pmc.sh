#!/bin/bash
# GPU box: LDS / issue counters of the flash-attention kernel alone (tools/kbench.py attn), one rocprofv3 --pmc pass per counter group.
#   bash tools/attn_pmc.sh <variant> <outdir>
set -e -o pipefail
V=${1:-0}
OUT=${2:-gpurun_out/attn_pmc_v$V}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python3 tools/kbench.py attn --images 32 --attn-variant $V"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2> $OUT/c.err
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for sub in "abc":
    cc = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True)
    if not cc: print(sub, "no counters", open(os.path.join(out, sub + ".err")).read()[-400:]); continue
    acc = defaultdict(list)
    for row in csv.DictReader(open(cc[0])):
        if "flash_attn" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "flash_attn" in r["Kernel_Name"]]
    d = sum(dur) / max(len(dur), 1)
    line = {k: sum(v) / len(v) for k, v in acc.items()}
    print(f"pass {sub}: launches {len(dur)} mean {d/1e3:.1f} us", " ".join(f"{k}={v:.4g}" for k, v in sorted(line.items())))
    if "GRBM_GUI_ACTIVE" in line: print(f"   clock = {line['GRBM_GUI_ACTIVE'] / 8 / d:.3f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)")
PY
