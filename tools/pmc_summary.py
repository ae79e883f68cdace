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
    for key in ("flash_attn_kernel", "gemm_kernel_v10", "gemm_kernel_v8", "gemm_kernel_v7", "gemm_kernel_v3", "gemm_kernel", "layernorm_kernel", "ln_prepare_kernel",
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

# ---- per template instantiation of the persistent GEMM: FETCH / WRITE per epilogue, next to the algorithmic bytes.
# EPI_RESID_SCALE_LN serves two shapes (out-proj K = 768, fc2 K = 3072): its launches are split at the midpoint of their fetch sizes.
import re
EPI_NAMES = {0: "EPI_STORE", 1: "EPI_GELU", 2: "EPI_HEADS", 3: "EPI_VT", 4: "EPI_RESID_SCALE", 5: "EPI_RESID_ADD", 6: "EPI_PATCH", 7: "EPI_STORE_F32",
             8: "EPI_QKV", 9: "EPI_RESID_SCALE_LN", 10: "EPI_QKV_LN", 11: "EPI_GELU_LN"}
M, D, F, KP, NP = 32 * 5376, 768, 3072, 640, 5376          # B = 32 images x 5376 padded token rows; bf16 operands (2 B), fp32 residual (4 B)
ALGO = {   # (read bytes, written bytes) per launch, formulas in DESIGN.md §4
    ("EPI_PATCH", None): (M * KP * 2 + NP * D * 4 + D * KP * 2, M * D * 4),
    ("EPI_QKV_LN", None): (M * D * 2 + M * 8 + 3 * D * D * 2, M * 3 * D * 2),
    ("EPI_GELU_LN", None): (M * D * 2 + M * 8 + F * D * 2, M * F * 2),
    ("EPI_RESID_SCALE_LN", "small"): (M * D * 2 + M * D * 4 + M * 4 + D * D * 2, M * D * 4 + M * D * 2 + M * 96),        # out-proj
    ("EPI_RESID_SCALE_LN", "large"): (M * F * 2 + M * D * 4 + M * 4 + D * F * 2, M * D * 4 + M * D * 2 + M * 96),        # fc2 feeding a fused LayerNorm
    ("EPI_RESID_SCALE", None): (M * F * 2 + M * D * 4 + D * F * 2, M * D * 4),                                            # fc2 of the last ViT / last align block
}


EPI_NAMES.update({12: "EPI_HEADS_LN", 13: "EPI_VT_LN", 14: "EPI_PATCH_LN"})
ALGO[("EPI_PATCH_LN", None)] = (M * KP * 2 + NP * D * 4 + D * KP * 2, M * D * 4 + M * D * 2 + M * 96)


def epi_of(name):
    """EPI template argument of a gemm_kernel_v8 / v10 instantiation: from the MANGLED name (tools/profile_round.sh runs the FETCH / WRITE passes with
    --mangled-kernels: _ZN2rz14gemm_kernel_v8IDF16bLi9ELb0ELb0ELin1EEEvNS_8GemmArgsE), else from a well-formed demangled one."""
    m = re.search(r"gemm_kernel_v(?:8|10)I(?:DF16b|DF16_|u6__bf16|Dh|f|[^L]*?)Li(\d+)E", name) or \
        re.search(r"gemm_kernel_v(?:8|10)<[^,>]+,\s*(?:\(rz::Epilogue\))?(\d+)", name)
    return EPI_NAMES.get(int(m.group(1))) if m else None


def is_bf16(name):
    return "bf16" in name or "DF16b" in name or "u6__bf16" in name


per = {}
for k in set(fetch) | set(write):
    e = epi_of(k)
    if not e or not is_bf16(k):
        continue
    f, w = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
    n = min(len(f), len(w))                     # same command, same dispatch order in both passes
    if not n:
        continue
    rows = [(2 * f[i] * 1024, w[i] * 1024) for i in range(n)]
    groups = {None: rows}
    if e == "EPI_RESID_SCALE_LN":
        mid = (min(r[0] for r in rows) + max(r[0] for r in rows)) / 2
        groups = {"small": [r for r in rows if r[0] < mid], "large": [r for r in rows if r[0] >= mid]}
    for tag, rr in groups.items():
        if not rr:
            continue
        rd, wr = sum(r[0] for r in rr) / len(rr), sum(r[1] for r in rr) / len(rr)
        a = ALGO.get((e, tag))
        per[e + ("" if tag is None else {"small": " (out-proj, K=768)", "large": " (fc2, K=3072)"}[tag])] = {
            "launches_sampled": len(rr), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "hbm_bytes": int(rd + wr),
            "algorithmic_read_bytes": a[0] if a else None, "algorithmic_write_bytes": a[1] if a else None,
            "ratio_to_algorithmic": round((rd + wr) / (a[0] + a[1]), 3) if a else None}
traffic["gemm_per_instantiation"] = per
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
