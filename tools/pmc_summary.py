"""Summarise the rocprofv3 passes of tools/profile_round.sh into gpurun_out/prof_<round>/summary/ (copied into profiles/<round>/):
  kernel_stats.csv            per-kernel time of the bench command (rocprofv3 --stats)
  hbm_traffic_pmc.json        HBM bytes per launch of the dominant kernels (read by bench.py for roofline.traffic)
  mfma_utilisation_pmc.json   MFMA busy / VALU active / wait fractions / shader clock per kernel
Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads (doubled); WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

rnd, out = sys.argv[1], sys.argv[2]
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def counters(sub):
    """kernel -> counter -> list of per-dispatch values; kernel -> list of durations (ns)"""
    path = find(sub, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    if path:
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = defaultdict(list)
    kt = find(sub, "*kernel_trace.csv")
    if kt:
        for row in csv.DictReader(open(kt)):
            dur[row["Kernel_Name"]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    return acc, dur


def short(name):
    for key in ("flash_attn_kernel", "gemm_kernel_v8", "gemm_kernel_v7", "gemm_kernel_v3", "gemm_kernel", "layernorm_kernel", "ln_prepare_kernel",
                "ln_finalize_kernel", "vlcabs_scores_kernel", "vlcabs_partial_kernel", "vlcabs_finalize_kernel", "vlcabs_kernel", "im2col_kernel"):
        if key in name:
            return key
    return None


stats = find("trace", "*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(summ, "kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "bench_default.json"):
    if os.path.exists(os.path.join(out, f)):
        shutil.copy(os.path.join(out, f), os.path.join(summ, f))

fetch, _ = counters("pmc_fetch")
write, _ = counters("pmc_write")
traffic = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                      "--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-events; summarised by tools/pmc_summary.py",
           "config": {"batch": 32, "image_side": 1024, "n_prompts": 14, "dtype": "bf16"},
           "correction": "MI355X_MICROARCH.md HBM section: counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced "
                         "reads -> doubled; WRITE_SIZE taken as is (16-byte-per-lane streaming stores).  hbm_bytes_per_launch = (2*FETCH + WRITE) * 1024, "
                         "mean over the launches of that kernel with at least half the maximum traffic (drops the small text-encoder launches).",
           "kernels": {}}
by = defaultdict(lambda: {"f": [], "w": []})
for k, c in fetch.items():
    if short(k):
        by[short(k)]["f"] += c.get("FETCH_SIZE", [])
for k, c in write.items():
    if short(k):
        by[short(k)]["w"] += c.get("WRITE_SIZE", [])
for key, v in by.items():
    f, w = v["f"], v["w"]
    if not f or not w:
        continue
    tot = [a for a in f]
    f = [a for a in f if a > 0.5 * max(f)] or f
    w = [a for a in w if a > 0.5 * max(w)] or w
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    traffic["kernels"][key] = {"launches_sampled": len(f), "FETCH_SIZE_KiB_raw": round(fm, 1), "WRITE_SIZE_KiB_raw": round(wm, 1),
                               "hbm_bytes_per_launch": int((2 * fm + wm) * 1024)}
json.dump(traffic, open(os.path.join(summ, "hbm_traffic_pmc.json"), "w"), indent=1)

sq, dur = counters("pmc_sq")
util = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU "
                   "--output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-events (B=32, 1024^2, bf16)",
        "notes": "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8); clock = GRBM_GUI_ACTIVE/8 / duration; SQ_WAVE_CYCLES, SQ_WAIT_* and "
                 "SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md); profiled passes clock lower than un-profiled ones",
        "kernels": {}}
agg = defaultdict(lambda: defaultdict(list))
for k, c in sq.items():
    if short(k):
        for name, vals in c.items():
            agg[short(k)][name] += vals
        agg[short(k)]["_dur"] += dur.get(k, [])
for key, c in agg.items():
    n = len(c.get("GRBM_GUI_ACTIVE", []))
    if not n:
        continue
    mean = lambda name: sum(c[name]) / max(1, len(c[name])) if c.get(name) else 0.0
    gui = mean("GRBM_GUI_ACTIVE") / 8.0
    d_ns = mean("_dur")
    util["kernels"][key] = {"launches": n, "mean_duration_us": round(d_ns / 1e3, 1), "shader_clock_GHz": round(gui / d_ns, 3) if d_ns else None,
                            "mfma_busy_frac_of_simd_cycles": round(mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui), 4) if gui else None,
                            "valu_active_frac": round(mean("SQ_ACTIVE_INST_VALU") / max(1.0, mean("SQ_WAVE_CYCLES")), 4),
                            "wave_cycles_waiting_frac": round(mean("SQ_WAIT_ANY") / max(1.0, mean("SQ_WAVE_CYCLES")), 4),
                            "wave_cycles_issue_stall_frac": round(mean("SQ_WAIT_INST_ANY") / max(1.0, mean("SQ_WAVE_CYCLES")), 4)}
json.dump(util, open(os.path.join(summ, "mfma_utilisation_pmc.json"), "w"), indent=1)
print(json.dumps({"traffic": traffic["kernels"], "util": util["kernels"]}, indent=1))
