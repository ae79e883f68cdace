"""GPU box: where does the fused-LayerNorm path lose accuracy on the outlier checkpoint?  Residual stream after k blocks
(depth-truncated models), bf16 fused / bf16 stand-alone against the fp32 kernels; max and rms error relative to the rms."""
import dataclasses, os, re, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from radzero_amd import _lib
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_pixels
from radzero_amd.weights import add_outlier_channels, make_state_dict
lib = _lib.load()
cfg = RadZeroConfig()
sd_full = add_outlier_channels(make_state_dict(cfg, 20260103), cfg)
px = torch.from_numpy(synthetic_pixels(2, 224, 1250)).cuda()
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
for depth in (1, 2, 3, 4, 6, 12):
    c = dataclasses.replace(cfg, vit_layers=depth, align_layers=0)
    sd = {k: v for k, v in sd_full.items() if not (re.match(r"vision_model\.encoder\.layer\.(\d+)\.", k) and int(re.match(r"vision_model\.encoder\.layer\.(\d+)\.", k).group(1)) >= depth) and not k.startswith("align_transformer.")}
    ref_m = RadZeroModel.from_state_dict(sd, c, torch_dtype=torch.float32, device="cuda:0").eval()
    ref = ref_m.forward_vision_model(px)["vision_tokens"].clone(); ref_m.close()
    m = RadZeroModel.from_state_dict(sd, c, torch_dtype=dt, device="cuda:0").eval()
    line = f"depth {depth:2d}:"
    for fused in (1, 0):
        lib.rz_set_option(b"ln_fused", fused)
        out = m.forward_vision_model(px)["vision_tokens"]
        d = (out - ref)
        rms = ref.pow(2).mean().sqrt()
        tok = int(d.abs().amax(-1).flatten().argmax())
        line += f"  fused={fused}: max {float(d.abs().max() / rms):.4f} rms {float(d.pow(2).mean().sqrt() / rms):.5f} (worst token {tok % 257} of image {tok // 257}, channel {int(d.abs().flatten().argmax()) % 768})"
    lib.rz_set_option(b"ln_fused", 1)
    m.close()
    print(line, flush=True)
