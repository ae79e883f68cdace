"""GPU box host: the CPU baseline (oracle, SDPA attention, 1024^2 x 14 cached prompts, fp32 — bench.py cpu_baseline's leg) at 16 / 32 / 64 / 128 threads,
recorded ONCE per round (the thread count of `cpu_baseline` must be a measurement, not a choice).  Writes
gpurun_out/r06/cpu_threads_sweep.{log,json}; bench.py runs its CPU leg at the fastest count found here.   python tools/cpu_threads_sweep.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from oracle.radzero_oracle import OracleModel  # noqa: E402  (the baseline leg: never on the product path)
from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.synthetic import synthetic_pixels, synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out", "r06")
os.makedirs(out_dir, exist_ok=True)
cfg = RadZeroConfig()
sd = make_state_dict(cfg, 20260103)
ids, mask = synthetic_prompts(14, 6, 10, 4321)
enc = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
om = OracleModel(sd, cfg, attn_impl="sdpa")
px = torch.from_numpy(synthetic_pixels(2, 1024, 1234))
phys = bench.physical_cores()
lines = [f"host: {bench.cpu_model_name()}, {os.cpu_count()} logical CPUs, {phys} physical cores, affinity mask {len(os.sched_getaffinity(0))}; "
         "oracle (oracle/radzero_oracle.py), SDPA attention, 2 x 1024^2 images x 14 cached prompts per pass, fp32; 1 warm-up + median of 3 passes"]
res = {}
with torch.no_grad():
    for n in (16, 32, 64, 128):
        if n > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(n)
        tf = om.text_features(enc, split_rows=False)
        om.compute_logits(px[:1, :, :512, :512], [enc], text_features=tf)
        ts = []
        for _ in range(3):
            t0 = time.time()
            om.compute_logits(px, [enc], text_features=tf)
            ts.append(time.time() - t0)
        med = sorted(ts)[1]
        res[str(n)] = round(2 / med, 5)
        lines.append(f"{n:4d} threads: {2 / med:.4f} images/s (passes {', '.join(f'{t:.2f}' for t in ts)} s)")
        print(lines[-1], flush=True)
best = max(res, key=lambda k: res[k])
lines.append(f"fastest: {best} threads, {res[best]} images/s")
open(os.path.join(out_dir, "cpu_threads_sweep.log"), "w").write("\n".join(lines) + "\n")
json.dump({"cpu": bench.cpu_model_name(), "logical_cpus": os.cpu_count(), "physical_cores": phys, "images_per_s_by_threads": res, "fastest_threads": int(best),
           "workload": "oracle SDPA, 2 x 1024^2 x 14 cached prompts per pass, fp32, median of 3"}, open(os.path.join(out_dir, "cpu_threads_sweep.json"), "w"), indent=1)
print("\n".join(lines))
