"""Single-image latency of compute_logits, eager and replayed as one hipGraph, bf16 and fp32, 224 / 518 / 1024 px (GPU box):
  python tools/latency_probe.py"""
import time, torch, numpy as np, sys
sys.path.insert(0, ".")
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.weights import make_state_dict
from radzero_amd.synthetic import synthetic_pixels, synthetic_prompts
cfg = RadZeroConfig()
sd = make_state_dict(cfg, 1)
for dt in (torch.bfloat16, torch.float32):
    m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=dt, device="cuda:0").eval()
    ids, mask = synthetic_prompts(1, 5, 9, 3)
    enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
    for S in (224, 518, 1024):
        px = torch.from_numpy(synthetic_pixels(1, S, 5)).cuda()
        for _ in range(3): m.compute_logits(px, [enc])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.compute_logits(px, [enc])
        torch.cuda.synchronize(); e = (time.perf_counter() - t0) / 20
        run = m.make_graphed(px.shape, [enc])
        for _ in range(3): run(px)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run(px)
        torch.cuda.synchronize(); g = (time.perf_counter() - t0) / 20
        print(f"{dt} S={S} B=1 T=1: eager {e*1e3:.3f} ms  hipGraph replay {g*1e3:.3f} ms")
    m.close()
