"""GPU box: per-GEMM time of the 128x128-kernel family inside the step, per tile geometry and ring depth, from rocprofv3 kernel traces.
   python tools/small_kernel_trace.py parse DIR   -> one table per shape: rows = (epilogue, grid-independent role), columns = forced configs
   (the runs themselves: tools/small_kernel_trace.sh).  Roles: EPI 9 (out-proj and fc2 share an instantiation) is split by duration."""
import csv
import glob
import math
import os
import re
import statistics
import sys
from collections import defaultdict

EPI = {0: "store", 1: "gelu(fc1)", 2: "heads(q|k)", 3: "vT(v)", 4: "resid_scale", 5: "resid_add", 6: "patch", 7: "store_f32", 8: "qkv merged", 9: "resid_scale_ln", 10: "gelu_ln(fc1)",
       11: "heads_ln(q|k)", 12: "vT_ln(v)", 13: "qkv_ln merged", 14: "patch_ln"}


def role(name):
    m = re.search(r"gemm_kernelI(\w+?)Li(\d+)E", name)
    if m:
        return f"small {m.group(1)} epi {m.group(2)}"
    if "gemm_kernel_v8" in name:
        return "persistent v8"
    if "gemm_kernel" in name:          # rocprofv3 half-demangles some instantiations: keep them apart by the text it printed
        return "small ? " + re.sub(r"[^A-Za-z0-9]+", "", name.split("gemm_kernel")[1])[:18]
    if "flash_attn" in name:
        return "attention"
    return None


def parse(path):
    rows = list(csv.DictReader(open(path)))
    by = defaultdict(list)
    for r in rows:
        k = role(r["Kernel_Name"])
        if k is None:
            continue
        # rows of the table: 64 x 64 output blocks = (M / 64)(N / 64), the same number whatever the tile geometry
        by[(k, int(r["Grid_Size_X"]) // 64 if k.startswith("small") else int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for (k, wgs), d in by.items():
        d = d[len(d) // 5:]          # skip warm-up
        lo, hi = min(d), max(d)
        if "epi 9" in k and hi > 1.8 * lo:          # out-proj (K = 768) and fc2 (K = 3072) share the instantiation
            thr = math.sqrt(lo * statistics.median(sorted(d)[-len(d) // 4:]))
            a = [x for x in d if x < thr]
            b = [x for x in d if x >= thr]
            out[(k + " short-K", wgs)] = (statistics.median(a), len(a))
            out[(k + " long-K", wgs)] = (statistics.median(b), len(b))
        else:
            out[(k, wgs)] = (statistics.median(d), len(d))
    return out


def main():
    root = sys.argv[2]
    shapes = sorted({os.path.basename(p).split("__")[0] for p in glob.glob(os.path.join(root, "*__*"))})
    for sh in shapes:
        cfgs = sorted({os.path.basename(p).split("__")[1] for p in glob.glob(os.path.join(root, sh + "__*"))}, key=int)
        tab = {}
        for c in cfgs:
            f = glob.glob(os.path.join(root, f"{sh}__{c}", "**", "*kernel_trace.csv"), recursive=True)
            if f:
                tab[c] = parse(f[0])
        print(f"== {sh}: median us per launch (workgroups); columns = gemm_small_tile " + " ".join(cfgs))
        totals = {c: 0.0 for c in tab}
        keys = sorted({k for t in tab.values() for k in t})
        for k in keys:
            cells = []
            for c in cfgs:
                v = tab.get(c, {}).get(k)
                if v is None:
                    cells.append("      -  ")
                    continue
                totals[c] += v[0] * v[1]
                cells.append(f"{v[0]:7.1f}x{v[1]:<3d}"[:11])
            print(f"{k[0]:40s} {k[1]:6d} blocks " + " ".join(cells))
        print(f"{'sum of medians x launches (ms)':52s} " + " ".join(f"{totals[c] / 1e3:13.2f}" for c in cfgs))


if __name__ == "__main__":
    main()
