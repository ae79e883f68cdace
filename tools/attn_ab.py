"""A/B of attention kernel variants (RZ_EXPERIMENTS=1 build): same inputs, outputs compared bit for bit with attn_variant 4, then timed.
  RZ_EXPERIMENTS=1 python3 tools/attn_ab.py 4 5 64 [--dtype f16] [--nv 5330]"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+", type=int)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--nv", type=int, default=5330)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--scale", type=float, default=0.5)
a = ap.parse_args()
lib = _lib.load()
B, H, nv = a.batch, 12, a.nv
npad = (nv + 127) // 128 * 128
td, code = {"bf16": (torch.bfloat16, 1), "f16": (torch.float16, 2)}[a.dtype]
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.manual_seed(0)
q = (torch.randn(B, H, npad, 64, device="cuda") * a.scale).to(td)
k = (torch.randn(B, H, npad, 64, device="cuda") * a.scale).to(td)
vt = torch.randn(B, H, 64, npad, device="cuda").to(td)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 4.0 * B * H * nv * nv * 64
ref = None
for rep in range(2):
    for v in a.variants:
        lib.rz_set_option(b"attn_variant", v)
        ctx = torch.zeros(B, npad, H * 64, device="cuda", dtype=td)
        f = lambda: lib.rz_flash_attention(code, P(q), P(k), P(vt), P(ctx), B, H, nv, npad, st)
        for _ in range(3):
            assert f() == 0, lib.rz_last_error()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out = ctx[:, :nv].clone()
        if ref is None:
            ref = out
        same = torch.equal(out.view(torch.int16), ref.view(torch.int16))
        print(f"variant {v:4d}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  bit-identical to first: {same}  max|d| {float((out.float() - ref.float()).abs().max()):.3g}", flush=True)
