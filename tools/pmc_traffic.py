"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench into
profiles/r01/hbm_traffic_pmc.json (read by bench.py for `roofline.traffic`).
  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv>
Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads (doubled here; calibrated on layernorm_kernel, whose traffic is known exactly)."""
import csv, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py "
                      "--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events; summarised by tools/pmc_traffic.py",
           "config": {"batch": 32, "image_side": 1024, "n_prompts": 14, "dtype": "bf16"},
           "correction": "MI355X_MICROARCH.md HBM section: counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide "
                         "coalesced reads -> doubled. WRITE_SIZE is exact for flash_attn_kernel (258048 KiB = the bf16 ctx rows it writes) but reads "
                         "2x for layernorm_kernel's stores (516096 KiB against 258048 KiB of bf16 rows actually written; the guide calls widths other "
                         "than 16 B per lane uncalibrated) -> write_factor 0.5 there. Calibration of the read side: layernorm_kernel reads "
                         "172032 x 768 fp32 = 516096 KiB, 2 x FETCH_SIZE = 516320 KiB.",
           "kernels": {}}
    for key, pat in (("layernorm_kernel", "layernorm_kernel"), ("flash_attn_kernel", "flash_attn_kernel"), ("gemm_kernel_v7", "gemm_kernel_v7")):
        f = [v for k, vs in fetch.items() if pat in k for v in vs]
        w = [v for k, vs in write.items() if pat in k for v in vs]
        if key == "layernorm_kernel":      # the big launches only (rows = B * Npad), not the text encoder's
            f = [v for v in f if v > 0.5 * max(f)]; w = [v for v in w if v > 0.5 * max(w)]
        if not f or not w:
            continue
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        wf = 0.5 if key == "layernorm_kernel" else 1.0
        out["kernels"][key] = {"launches_sampled": len(f), "FETCH_SIZE_KiB_raw": round(fm, 1), "WRITE_SIZE_KiB_raw": round(wm, 1), "write_factor": wf,
                               "hbm_bytes_per_launch": int((2 * fm + wf * wm) * 1024)}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic_pmc.json"), "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
