"""Summarise the rocprofv3 --pmc pass of tools/attn_ablate.py: per flash_attn_kernel instantiation (ABL mask) the mean launch duration,
shader clock (GRBM_GUI_ACTIVE / duration), MFMA-busy fraction, VALU-active and wait fractions of wave cycles.
  python3 tools/attn_ablate_pmc.py gpurun_out/abl/pmc"""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"flash_attn_kernelI(\w+?), 4, (\d), (true|false), (true|false), (\d+)>", r["Kernel_Name"]) or \
            re.search(r"flash_attn_kernelIDF16bLi4ELi(\d)ELb(\d)ELb(\d)ELi(\d+)E", r["Kernel_Name"])
        if not m:
            continue
        key = (int(m.groups()[-1]), int(m.groups()[-4]) if len(m.groups()) == 5 else int(m.group(1)))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[key]["ns:" + r["Dispatch_Id"]] = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])]
names = {1: "noexp", 2: "noPV", 4: "noQK", 8: "nolds", 16: "nodma", 32: "nobar", 64: "quarterlds"}
print(f"{'ABL':>4} {'QT':>2} {'what':28s} {'ms':>7} {'GHz':>5} {'mfma_busy':>9} {'busy*GHz':>8} {'valu/wave':>9} {'wait_any':>8} {'wait_inst':>9}")
for (abl, qt), c in sorted(rows.items()):
    mean = lambda k: sum(c[k]) / max(len(c[k]), 1)
    ns = [v[0] for k, v in c.items() if k.startswith("ns:")]
    dur = sum(ns) / len(ns)
    clk = mean("GRBM_GUI_ACTIVE") / 8.0 / dur    # GRBM_GUI_ACTIVE sums the 8 XCDs; cycles per ns = GHz
    # MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), as tools/pmc_summary.py
    busy = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (mean("GRBM_GUI_ACTIVE") / 8.0 * 1024)
    wc = mean("SQ_WAVE_CYCLES")
    label = "+".join(v for b, v in names.items() if abl & b) or "full"
    print(f"{abl:4d} {qt:2d} {label:28s} {dur / 1e6:7.3f} {clk:5.2f} {busy:9.3f} {busy * clk:8.3f} {mean('SQ_ACTIVE_INST_VALU') / wc:9.3f} "
          f"{mean('SQ_WAIT_ANY') / wc:8.3f} {mean('SQ_WAIT_INST_ANY') / wc:9.3f}")
