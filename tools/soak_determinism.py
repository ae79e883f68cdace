"""Soak: the full B=32 x 1024^2 bf16 step repeated N times on the same input must give the same bits every time (LDS-DMA / barrier
races show up as rare differing runs).  python tools/soak_determinism.py [iterations] [dtype] [batch] [side] [fresh-text]
(round 6: batch / side select the small-shape kernels — the 128 x 128 MX kernel of the fp32 mode —, "fresh-text" disables the prompt cache so that the text
encoder — three-plane GEMMs with its own predicated guard in the fp32 mode — runs in every iteration)"""
import sys, time
import torch
sys.path.insert(0, ".")
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_prompts
from radzero_amd.weights import make_state_dict

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
fresh = len(sys.argv) > 5 and sys.argv[5] == "fresh-text"
cfg = RadZeroConfig()
m = RadZeroModel.from_state_dict(make_state_dict(cfg, 1), cfg, torch_dtype=dtype, device="cuda:0").eval()
g = torch.Generator(device="cuda").manual_seed(99)
px = torch.randn((B, 3, S, S), generator=g, device="cuda")
if fresh:
    m.text_cache_enabled = False
ids, mask = synthetic_prompts(14, 6, 10, 4321)
enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
out = m.compute_logits(px, [enc])
ref_l, ref_s = out["logits"].clone(), out["similarity_scores"].clone()
bad = 0
t0 = time.time()
for i in range(iters):
    out = m.compute_logits(px, [enc])
    if not (torch.equal(out["logits"], ref_l) and torch.equal(out["similarity_scores"], ref_s)):
        bad += 1
        print(f"iteration {i}: differs, max |d score| = {(out['similarity_scores'] - ref_s).abs().max().item():.3e}", flush=True)
    if i % 50 == 49:
        print(f"{i + 1} iterations, {bad} differing, {time.time() - t0:.0f} s", flush=True)
print(f"soak {dtype} B={B} {S}^2{' fresh text every iteration' if fresh else ''}: {iters} iterations, {bad} differing runs, guard re-runs {m.guard_reruns() if dtype == torch.float32 else 'n/a'}")
sys.exit(1 if bad else 0)
