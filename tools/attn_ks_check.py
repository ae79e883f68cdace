"""Numerics of an experimental attention variant against the default kernel (RZ_EXPERIMENTS=1 build), one (image, head):
  RZ_EXPERIMENTS=1 python3 tools/attn_ks_check.py <attn_variant> <n_valid>    (128 = the key-split kernel; per 16-row block maxima are printed)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
lib = _lib.load()
B, H, nv = 1, 1, int(sys.argv[2]) if len(sys.argv) > 2 else 512
npad = (nv + 255) // 256 * 256
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.manual_seed(0)
q = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
k = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
vt = torch.randn(B, H, 64, npad, device="cuda").bfloat16()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = {}
for v in (4, int(sys.argv[1])):
    lib.rz_set_option(b"attn_variant", v)
    ctx = torch.zeros(B, npad, H * 64, device="cuda", dtype=torch.bfloat16)
    assert lib.rz_flash_attention(1, P(q), P(k), P(vt), P(ctx), B, H, nv, npad, st) == 0
    torch.cuda.synchronize()
    outs[v] = ctx[0, :nv].float().cpu()
a, b = outs[4], outs[int(sys.argv[1])]
print("nonfinite", int((~torch.isfinite(b)).sum()), "max|d|", float((a - b).abs().nan_to_num(1e9).max()))
d = (a - b).abs().nan_to_num(1e9)
rows = d.max(1).values
for r0 in range(0, nv, 16):
    print(r0, ["%.3g" % float(x) for x in d[r0:r0 + 16].max(0).values.view(4, 16).max(1).values], "rowmax %.3g" % float(rows[r0:r0 + 16].max()))
