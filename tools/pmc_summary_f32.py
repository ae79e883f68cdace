"""HBM bytes per launch of the fp32 (1e-3) mode's kernels from the two counter passes of tools/profile_round.sh (pmc_fetch_f32 / pmc_write_f32:
`bench.py --dtype f32`), corrected as tools/pmc_summary.py does (MI355X_MICROARCH.md: KiB; FETCH_SIZE x 2 on gfx950; WRITE_SIZE as is)
-> <out>/summary/hbm_traffic_pmc_f32.json, read by bench.py for the fp32 lines' `roofline.traffic`.   python3 tools/pmc_summary_f32.py <out>"""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)


def counters(sub, name):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def short(name):
    for key in ("flash_attn_split_kernel", "flash_attn_kernel", "gemm_kernel_v8", "gemm_kernel_v7", "gemm_kernel_v3", "gemm_kernel", "layernorm_split_mx_kernel", "layernorm_split3_kernel",
                "layernorm_kernel", "split_mx_rows_kernel", "vlcabs_finalize_kernel", "vlcabs_kernel", "im2col_kernel", "copy_tokens_kernel"):
        if key in name:
            return key
    return None


fetch, write = counters("pmc_fetch_f32", "FETCH_SIZE"), counters("pmc_write_f32", "WRITE_SIZE")
by = defaultdict(lambda: {"f": [], "w": []})
for k, v in fetch.items():
    if short(k):
        by[short(k)]["f"] += v
for k, v in write.items():
    if short(k):
        by[short(k)]["w"] += v
rec = {"command": "rocprofv3 --kernel-trace --mangled-kernels --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --dtype f32 "
                  "--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-events; summarised by tools/pmc_summary_f32.py",
       "config": {"batch": 32, "image_side": 1024, "n_prompts": 14, "dtype": "f32"},
       "correction": "MI355X_MICROARCH.md HBM section: counters in KiB; FETCH_SIZE doubled on gfx950; WRITE_SIZE as is; mean over the launches with at least half the kernel's maximum "
                     "traffic (drops the text encoder's small launches and the overflow guard's predicated, empty second-pass launches)",
       "kernels": {}}
for key, v in by.items():
    f, w = v["f"], v["w"]
    if not f or not w:
        continue
    f = [a for a in f if a > 0.5 * max(f)] or f
    w = [a for a in w if a > 0.5 * max(w)] or w
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    rec["kernels"][key] = {"launches_sampled": len(f), "FETCH_SIZE_KiB_raw": round(fm, 1), "WRITE_SIZE_KiB_raw": round(wm, 1), "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
n, d = 5330, 768
rec["algorithmic_bytes_note"] = {"flash_attn_split_kernel": int(32 * n * d * 4 * 4),
                                 "how": "B x N x 768 x 4 bytes per operand element (q | k hi + lo f16 planes, V^T hi f16 + e4m3 pair plane, ctx in the MX form) x 4 tensors"}
json.dump(rec, open(os.path.join(summ, "hbm_traffic_pmc_f32.json"), "w"), indent=1)
print(json.dumps(rec["kernels"], indent=1))

# ---- MFMA busy / VALU active / wait fractions / shader clock of the fp32 mode's kernels (pass pmc_sq_f32), as tools/pmc_summary.py computes them for the bf16 step
sq = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for path in glob.glob(os.path.join(out, "pmc_sq_f32", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        if k:
            sq[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for path in glob.glob(os.path.join(out, "pmc_sq_f32", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = short(row["Kernel_Name"])
        if k:
            dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
util = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -- "
                   "python3 bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-events (B=32, 1024^2, fp32 mode, default options)",
        "notes": "as profiles/*/mfma_utilisation_pmc.json; launches with less than half the kernel's longest duration are dropped (the text encoder's small launches, the guard's empty ones)",
        "kernels": {}}
for key, c in sq.items():
    d = dur.get(key, [])
    if not c.get("GRBM_GUI_ACTIVE") or not d:
        continue
    keep = [i for i, v in enumerate(d) if v > 0.5 * max(d)] if len(d) == len(c["GRBM_GUI_ACTIVE"]) else list(range(len(c["GRBM_GUI_ACTIVE"])))
    mean = lambda name: (sum(c[name][i] for i in keep if i < len(c[name])) / max(1, len([i for i in keep if i < len(c[name])]))) if c.get(name) else 0.0
    gui = mean("GRBM_GUI_ACTIVE") / 8.0
    d_ns = sum(d[i] for i in keep if i < len(d)) / max(1, len([i for i in keep if i < len(d)]))
    util["kernels"][key] = {"launches": len(keep), "mean_duration_us": round(d_ns / 1e3, 1), "shader_clock_GHz": round(gui / d_ns, 3) if d_ns else None,
                            "mfma_busy_frac_of_simd_cycles": round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui), 4) if gui else None,
                            "valu_active_frac": round(mean("SQ_ACTIVE_INST_VALU") / max(1.0, mean("SQ_WAVE_CYCLES")), 4),
                            "wave_cycles_waiting_frac": round(mean("SQ_WAIT_ANY") / max(1.0, mean("SQ_WAVE_CYCLES")), 4),
                            "wave_cycles_issue_stall_frac": round(mean("SQ_WAIT_INST_ANY") / max(1.0, mean("SQ_WAVE_CYCLES")), 4)}
if util["kernels"]:
    json.dump(util, open(os.path.join(summ, "mfma_utilisation_pmc_f32.json"), "w"), indent=1)
    print(json.dumps(util["kernels"], indent=1))
