"""Per-instantiation clock / MFMA-busy summary of a rocprofv3 --pmc pass over tools/attn_ks_ablate.py or bench.py --attn-variant 128:
  python3 tools/ks_pmc.py <rocprof output dir>"""
import csv, glob, re, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"flash_attn_ks_kernel<[^>]*?(\d+)>", r["Kernel_Name"])
        key = ("ks", int(m.group(1))) if m else (("base", 0) if "flash_attn_kernel" in r["Kernel_Name"] else None)
        if key is None: continue
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[key]["ns:" + r["Dispatch_Id"]] = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])]
print(f"{'kernel':>10} {'ms':>7} {'GHz':>5} {'mfma_busy':>9} {'busy*GHz':>8} {'valu/wave':>9} {'wait_any':>8} {'wait_inst':>9}")
for key, c in sorted(rows.items()):
    mean = lambda k: sum(c[k]) / max(len(c[k]), 1)
    ns = [v[0] for k, v in c.items() if k.startswith("ns:")]
    dur = sum(ns) / len(ns)
    clk = mean("GRBM_GUI_ACTIVE") / 8.0 / dur
    busy = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (mean("GRBM_GUI_ACTIVE") / 8.0 * 1024)
    wc = mean("SQ_WAVE_CYCLES")
    print(f"{key[0] + str(key[1]):>10} {dur / 1e6:7.3f} {clk:5.2f} {busy:9.3f} {busy * clk:8.3f} {mean('SQ_ACTIVE_INST_VALU') / wc:9.3f} {mean('SQ_WAIT_ANY') / wc:8.3f} {mean('SQ_WAIT_INST_ANY') / wc:9.3f}")
