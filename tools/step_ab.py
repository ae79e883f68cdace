"""In-step A/B of kernel options (a variant is judged INSIDE the step first): ONE process, ONE model, the headline workload
(B images of side^2, T cached prompts), option sets applied with rz_set_model_option and timed in interleaved rounds
(cdna_hip_programming.md rule 24).  Per configuration: ms per step (median / min over rounds) and the per-family HIP-event times.

  python3 tools/step_ab.py "gemm_variant=8" "gemm_variant=12" "gemm_variant=12,gemm_raster=8" [--rounds 5 --steps 6] [--dtype bf16]
      [--batch 32 --side 1024 --prompts 14] [--check]      (--check: vision tokens of every configuration bit-identical to the first)
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.modeling import RadZeroModel  # noqa: E402
from radzero_amd.synthetic import synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="+", help='comma-separated option=value lists, e.g. "gemm_variant=12,gemm_raster=8"')
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--prompts", type=int, default=14)
ap.add_argument("--check", action="store_true")
ap.add_argument("--json", default=None, help="also write the table to this file")
a = ap.parse_args()

dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
cfg = RadZeroConfig()
dev = torch.device("cuda", 0)
model = RadZeroModel.from_state_dict(make_state_dict(cfg, 20260103), cfg, torch_dtype=dt, device=dev).eval()
g = torch.Generator(device=dev).manual_seed(1234)
px = torch.randn((a.batch, 3, a.side, a.side), generator=g, device=dev)
ids, mask = synthetic_prompts(a.prompts, 6, 10, 4321)
enc = {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev)}
tf = model.forward_text_model(enc)["text_features_wo_l2_norm"]


def parse(c):
    return [(kv.split("=")[0], int(kv.split("=")[1])) for kv in c.split(",") if kv]


ALL = sorted({k for c in a.configs for k, _ in parse(c)})


def apply(c):
    for k in ALL:
        model.set_model_option(k, None)
    for k, v in parse(c):
        model.set_model_option(k, v)


ref = None
for c in a.configs:                      # warm-up + optional bit-identity check
    apply(c)
    for _ in range(2):
        out = model.compute_logits(px, [enc], text_features=tf)
    torch.cuda.synchronize()
    if a.check:
        tok = model.forward_vision_model(px)["vision_tokens"]
        if ref is None:
            ref = tok.clone()
        print(f"[check] {c}: bit-identical to first = {bool(torch.equal(tok, ref))}  finite = {bool(torch.isfinite(tok).all())}", flush=True)

ms = {c: [] for c in a.configs}
fam = {c: {} for c in a.configs}
for r in range(a.rounds):
    for c in a.configs:
        apply(c)
        model.compute_logits(px, [enc], text_features=tf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model.compute_logits(px, [enc], text_features=tf)
        torch.cuda.synchronize()
        ms[c].append((time.perf_counter() - t0) / a.steps * 1e3)
for c in a.configs:                      # per-family times: two extra steps with every family's HIP events
    apply(c)
    model.compute_logits(px, [enc], text_features=tf)
    torch.cuda.synchronize()
    model.profile(True)
    for _ in range(2):
        model.compute_logits(px, [enc], text_features=tf)
    torch.cuda.synchronize()
    p = model.profile_read()
    model.profile(False)
    fam[c] = {k: round(v["ms"] / 2, 3) for k, v in p.items()}

rows = []
for c in a.configs:
    med, mn = statistics.median(ms[c]), min(ms[c])
    rows.append({"config": c, "ms_per_step_median": round(med, 3), "ms_per_step_min": round(mn, 3),
                 "images_per_s_median": round(a.batch / med * 1e3, 1), "rounds": [round(x, 2) for x in ms[c]], "family_ms_per_step": fam[c]})
    print(f"{c:40s} median {med:8.3f} ms  min {mn:8.3f} ms  {a.batch / med * 1e3:7.1f} images/s  families {fam[c]}", flush=True)
if a.json:
    json.dump({"workload": f"B={a.batch} {a.side}^2 T={a.prompts} {a.dtype}", "rounds": a.rounds, "steps": a.steps, "rows": rows}, open(a.json, "w"), indent=1)
model.close()
