"""GPU box: fused vs stand-alone LayerNorm, error statistics (max and rms) against the reference goldens on the benign (G2)
and the outlier-channel (G8) checkpoints, bf16 and fp16."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import load_golden
from radzero_amd import _lib
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_pixels
from radzero_amd.weights import add_outlier_channels, make_state_dict
lib = _lib.load()
cfg = RadZeroConfig()
base = make_state_dict(cfg, 20260103)
for ck, gname, sd in (("benign", "g2_s224_b2_t3", base), ("outlier", "g8_outlier_s224_b2_t3", add_outlier_channels(base, cfg))):
    g = load_golden(gname)
    for dt in (torch.bfloat16, torch.float16):
        m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=dt, device="cuda:0").eval()
        px = torch.from_numpy(synthetic_pixels(int(g["batch"]), int(g["side"]), int(g["px_seed"]))).cuda()
        enc = {"input_ids": torch.from_numpy(g["input_ids"]).cuda(), "attention_mask": torch.from_numpy(g["attention_mask"]).cuda()}
        for fused in (1, 0):
            lib.rz_set_option(b"ln_fused", fused)
            out = m.compute_logits(px, [enc])
            d = out["similarity_scores"].cpu().numpy() - g["similarity_scores"]
            dl = out["logits"].cpu().numpy() - g["logits"]
            print(f"{ck:8s} {str(dt):15s} ln_fused={fused}: scores max {np.abs(d).max():.4f} rms {np.sqrt((d * d).mean()):.5f}   logits max {np.abs(dl).max():.4f}")
        lib.rz_set_option(b"ln_fused", 1)
        m.close()
