"""fp32 mode against the reference-run goldens (tests/golden): max |d similarity_scores| and |d logits| with the hi/lo-split f16 paths
(attention, GEMMs) on and off, and against the outlier-channel checkpoint.  python tools/fp32_split_accuracy.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import GOLDEN_CASES, load_golden
from radzero_amd import _lib
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_pixels
from radzero_amd.weights import make_state_dict

lib = _lib.load()
cfg = RadZeroConfig()
sd = make_state_dict(cfg, 20260103)      # the checkpoint of tests/conftest.py (the goldens were produced from it)
m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=torch.float32, device="cuda:0").eval()
m.set_model_option("pad_rows", 256)      # every padded token count a multiple of 256, so that the MX form (gemm_f32_mx = 2) applies to every fixture
for attn, gemm, mx in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 2)):
    lib.rz_set_option(b"attn_f32_split", attn); lib.rz_set_option(b"gemm_f32_split", gemm); lib.rz_set_option(b"gemm_f32_mx", mx)
    worst = []
    for name in GOLDEN_CASES:
        g = load_golden(name)
        px = torch.from_numpy(synthetic_pixels(int(g["batch"]), int(g["side"]), int(g["px_seed"]))).cuda()
        enc = {"input_ids": torch.from_numpy(g["input_ids"]).cuda(), "attention_mask": torch.from_numpy(g["attention_mask"]).cuda()}
        out = m.compute_logits(px, [enc])
        es = float(np.abs(out["similarity_scores"].cpu().numpy() - g["similarity_scores"]).max())
        el = float(np.abs(out["logits"].cpu().numpy() - g["logits"]).max())
        worst.append((name, es, el))
    print(f"attn_f32_split={attn} gemm_f32_split={gemm} gemm_f32_mx={mx}: " + "  ".join(f"{n.split('_')[0]}:{es:.1e}/{el:.1e}" for n, es, el in worst))
    print(f"    max over the 8 fixtures: scores {max(w[1] for w in worst):.2e}  logits {max(w[2] for w in worst):.2e}")
lib.rz_set_option(b"attn_f32_split", 1); lib.rz_set_option(b"gemm_f32_split", 1); lib.rz_set_option(b"gemm_f32_mx", 1)
m.close()
