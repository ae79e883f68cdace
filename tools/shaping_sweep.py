"""GPU box: images per second of one forward size against the batch-shaping model (radzero_amd/shaping.py) — 518^2 around 64 / 48 / 32 images, 224^2 x 256.
   python tools/shaping_sweep.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.modeling import RadZeroModel  # noqa: E402
from radzero_amd.synthetic import synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

cfg = RadZeroConfig()
sd = make_state_dict(cfg, 20260103)
m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=torch.bfloat16, device="cuda:0").eval()
ids, mask = synthetic_prompts(14, 6, 10, 4321)
enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
tf = m.forward_text_model(enc)["text_features_wo_l2_norm"]
for side, sizes in ((518, (64, 62, 60, 48, 46, 44, 32, 30, 28, 96, 92, 128, 124)), (224, (256, 248, 240)), (1024, (32, 30, 16))):
    n = (side // 14) ** 2 + 1
    for b in sizes:
        px = torch.randn((b, 3, side, side), device="cuda")
        for _ in range(3):
            m.compute_logits(px, [enc], text_features=tf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 12
        for _ in range(reps):
            m.compute_logits(px, [enc], text_features=tf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{side}^2 x {b:3d}: {b / dt:8.1f} images/s  {dt * 1e3:7.3f} ms   model cost/image {m.gemm_tile_cost(b, n) if hasattr(m, 'gemm_tile_cost') else float('nan'):.4g}  preferred_batch({b}) = {m.preferred_batch(b, side, side)}", flush=True)
        del px
m.close()
