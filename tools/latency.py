"""Single-image latency of compute_logits (GPU box): python tools/latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_prompts
from radzero_amd.weights import make_state_dict
cfg = RadZeroConfig(); sd = make_state_dict(cfg, 20260103)
for dt in (torch.bfloat16,):
    m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=dt, device="cuda:0").eval()
    ids, mask = synthetic_prompts(14, 6, 10, 1)
    enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
    for S, B in ((224, 1), (518, 1), (1024, 1), (1024, 4)):
        px = torch.randn(B, 3, S, S, device="cuda")
        for _ in range(3): m.compute_logits(px, [enc])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): out = m.compute_logits(px, [enc])
        torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t0) / n * 1e3
        run = m.make_graphed(px.shape, [enc])
        for _ in range(3): run(px)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = run(px)
        torch.cuda.synchronize(); g_ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{dt} S={S} B={B}: eager {dt_ms:.3f} ms per call ({B / dt_ms * 1e3:.1f} images/s) | hipGraph replay {g_ms:.3f} ms ({B / g_ms * 1e3:.1f} images/s)")
    m.close()
