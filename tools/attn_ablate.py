"""Attention timing-ablation probe (RZ_EXPERIMENTS=1 build; flash_attn_kernel's ABL template parameter, attn_variant 1000 + mask): ONE
ViT-B attention launch of the headline shape (32 images x 12 heads, 5330 tokens padded to 5376, bf16) run repeatedly with pieces of the
hot loop removed, for rocprofv3 --pmc to attribute clock and MFMA-busy to each.
  RZ_EXPERIMENTS=1 python3 tools/attn_ablate.py 0 1 2 4 8 16 32 ...
mask bits: 1 no exponentials, 2 no P V / row-sum MFMAs, 4 no score MFMAs, 8 no LDS fragment reads, 16 no K / V staging, 32 no barriers.
Results of an ablated launch are WRONG by construction; only the timing matters."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
assert os.environ.get("RZ_EXPERIMENTS") == "1", "needs the experiments build"
lib = _lib.load()
B, H, nv, npad = 32, 12, 5376, 5376      # no ragged tile: the masked keys' -inf would poison the no-exponential legs
P = lambda t: ctypes.c_void_p(t.data_ptr())
q = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
k = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
vt = torch.randn(B, H, 64, npad, device="cuda").bfloat16()
ctx = torch.empty(B, npad, H * 64, device="cuda", dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 4.0 * B * H * nv * nv * 64
names = {1: "noexp", 2: "noPV", 4: "noQK", 8: "nolds", 16: "nodma", 32: "nobar", 64: "quarterlds"}
for arg in sys.argv[1:] or ["0"]:
    if arg.startswith("v"):          # a plain attn_variant (v4 = product kernel, v64 = 64 query rows per wave, v417 = tracked maximum ...)
        mask = -1
        lib.rz_set_option(b"attn_variant", int(arg[1:]))
    else:
        mask = int(arg.lstrip("w"))          # "w<mask>": the 64-query-rows-per-wave shape (QT = 4)
        lib.rz_set_option(b"attn_variant", (2000 if arg.startswith("w") else 1000) + mask)
    f = lambda: lib.rz_flash_attention(1, P(q), P(k), P(vt), P(ctx), B, H, nv, npad, st)
    for _ in range(3):
        assert f() == 0, lib.rz_last_error()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    label = f"attn_variant {arg[1:]}" if mask < 0 else ("+".join(v for b, v in names.items() if mask & b) or "full (ABL = 0)") + (" [64 rows/wave]" if arg.startswith("w") else "")
    print(f"abl {mask:3d} {label:44s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s-equivalent", flush=True)
