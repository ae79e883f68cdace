"""VL-CABS head alone (rz_vlcabs: vlcabs_kernel + finalize) on the token tensor a reduced-depth vision forward leaves behind:
ms per call by HIP events around 20 calls, and a digest of scores / logits so that two library builds can be compared bit for bit.
  python tools/kvlcabs.py [--batch 32 --side 1024 --prompts 14]      (RZ_LIB_PATH=... selects another build)"""
import argparse
import ctypes
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from radzero_amd import _lib  # noqa: E402
from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.modeling import RadZeroModel  # noqa: E402
from radzero_amd.synthetic import synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--side", type=int, default=1024)
ap.add_argument("--prompts", type=int, default=14)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
cfg = RadZeroConfig(vit_layers=1, align_layers=1, text_layers=1)
dev = torch.device("cuda", 0)
m = RadZeroModel.from_state_dict(make_state_dict(cfg, 11), cfg, torch_dtype=torch.bfloat16, device=dev).eval()
px = torch.randn((a.batch, 3, a.side, a.side), generator=torch.Generator(device=dev).manual_seed(3), device=dev)
ids, mask = synthetic_prompts(a.prompts, 6, 12, 9)
enc = {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev)}
tf = m.forward_text_model(enc)["text_features_wo_l2_norm"]
out = m.compute_logits(px, [enc], text_features=tf)
torch.cuda.synchronize()
dig = hashlib.sha256(out["t2i_attn_weights"][0].cpu().numpy().tobytes() + out["logits"].cpu().numpy().tobytes()).hexdigest()[:16]
b, t = a.batch, a.prompts
n = out["t2i_attn_weights"][0].shape[-1]
scores = torch.empty((b, t, n), device=dev)
t2i = torch.empty((t, b), device=dev)
logits = torch.empty((b, t), device=dev)
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
lib = _lib.load()
for _ in range(3):
    _lib.check(lib.rz_vlcabs(m._h, P(tf), t, b, P(scores), P(t2i), P(logits), st))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    _lib.check(lib.rz_vlcabs(m._h, P(tf), t, b, P(scores), P(t2i), P(logits), st))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
tok_bytes = b * n * 768 * 4
print(f"vlcabs B={b} S={a.side} T={t} N={n}: {ms * 1e3:.1f} us per call, {tok_bytes / ms / 1e6:.0f} GB/s of token reads (+ {b * t * n * 4 / 1e6:.1f} MB of scores), digest {dig}  [{os.environ.get('RZ_LIB_PATH', 'default library')}]")
m.close()
