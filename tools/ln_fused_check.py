"""GPU box: the fused-LayerNorm path against the stand-alone-LayerNorm path and the reference golden (G7, 1024^2) inside a
full batch, plus timing of both.  python tools/ln_fused_check.py [dtype=bf16|f16] [batch]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from radzero_amd import _lib
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_pixels
from radzero_amd.weights import add_outlier_channels, make_state_dict
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = _lib.load()
cfg = RadZeroConfig()
g = load_golden("g7_s1024_b1_t14")
sd = make_state_dict(cfg, 20260103)
m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=dt, device="cuda:0").eval()
gen = torch.Generator(device="cuda").manual_seed(99)
px = torch.randn((B, 3, 1024, 1024), generator=gen, device="cuda")
px[min(5, B - 1)] = torch.from_numpy(synthetic_pixels(1, 1024, int(g["px_seed"]))[0]).cuda()
enc = {"input_ids": torch.from_numpy(g["input_ids"]).cuda(), "attention_mask": torch.from_numpy(g["attention_mask"]).cuda()}
res = {}
for fused in (1, 0):
    lib.rz_set_option(b"ln_fused", fused)
    out = m.compute_logits(px, [enc])
    torch.cuda.synchronize()
    sim, lg = out["similarity_scores"].clone(), out["logits"].clone()
    i = min(5, B - 1)
    e_s = np.abs(sim[i:i + 1].cpu().numpy() - g["similarity_scores"]).max()
    e_l = np.abs(lg[i].cpu().numpy() - g["logits"]).max()
    for _ in range(2):
        m.compute_logits(px, [enc])
    torch.cuda.synchronize()
    m.profile(True)
    t0 = time.perf_counter()
    for _ in range(8):
        m.compute_logits(px, [enc])
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / 8
    prof = m.profile_read(); m.profile(False)
    res[fused] = (sim, lg)
    print(f"ln_fused={fused} {dt} B={B}: golden image max|dscores|={e_s:.4f} max|dlogits|={e_l:.4f}  {B / dtm:.1f} images/s  " +
          " ".join(f"{k}={v['ms'] / 8:.2f}ms/{v['launches'] // 8}" for k, v in prof.items()), flush=True)
print(f"fused vs stand-alone: max|dscores|={(res[1][0] - res[0][0]).abs().max().item():.4f} max|dlogits|={(res[1][1] - res[0][1]).abs().max().item():.4f}")
lib.rz_set_option(b"ln_fused", 1)
