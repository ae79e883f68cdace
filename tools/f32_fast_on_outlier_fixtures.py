import sys, numpy as np, torch
sys.path.insert(0, ".")
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.synthetic import synthetic_pixels
from radzero_amd.weights import make_state_dict, add_outlier_channels
cfg = RadZeroConfig()
sd = add_outlier_channels(make_state_dict(cfg, 20260103), cfg)
m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=torch.float32, device="cuda:0").eval()
for name in ("g8_outlier_s224_b2_t3", "g15_outlier_s518_b2_t14", "g14_outlier_s1024_b1_t14"):
    g = dict(np.load(f"tests/golden/{name}.npz"))
    px = torch.from_numpy(synthetic_pixels(int(g["batch"]), int(g["side"]), int(g["px_seed"]))).cuda()
    enc = {"input_ids": torch.from_numpy(g["input_ids"]).cuda(), "attention_mask": torch.from_numpy(g["attention_mask"]).cuda()}
    for level in ("high", "fast"):
        m.set_f32_precision(level)
        out = m.compute_logits(px, [enc])
        e_s = float(np.abs(out["similarity_scores"].cpu().numpy() - g["similarity_scores"]).max())
        e_l = float(np.abs(np.atleast_2d(out["logits"].cpu().numpy()) - np.atleast_2d(g["logits"])).max())
        print(f"{name} f32_precision {level}: max|dscores| {e_s:.2e} max|dlogits| {e_l:.2e} (form {m.get_model_option('last_f32_form')})", flush=True)
m.close()
