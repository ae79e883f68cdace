"""Timing of the key-split attention kernel's generated loop and its ablated texts (RZ_EXPERIMENTS=1 build with the .inc generated under
RZ_KS_ABLATIONS=1): attn_variant 128 = the kernel, 129.. = novalu, nodma, nords, nobar, nop1mfma, nopvmfma, mfmaonly (results WRONG)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
lib = _lib.load()
nvs = [int(x) for x in os.environ.get('KS_NV', '5376').split(',')]
B, H = 32, 12
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {4: "default kernel", 64: "64 rows per wave", 128: "key split", 129: "ks novalu", 130: "ks nodma", 131: "ks nords", 132: "ks nobar", 133: "ks exp->mov", 134: "ks nocvt", 135: "ks mfma only"}
data = {}
for nv in nvs:
    npad = nv
    q = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
    k = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
    vt = torch.randn(B, H, 64, npad, device="cuda").bfloat16()
    ctx = torch.empty(B, npad, H * 64, device="cuda", dtype=torch.bfloat16)
    data[nv] = (q, k, vt, ctx)
for arg in sys.argv[1:] or ["4", "128"]:
    v = int(arg)
    lib.rz_set_option(b"attn_variant", v)
    per = {}
    for nv in nvs:
        q, k, vt, ctx = data[nv]
        f = lambda: lib.rz_flash_attention(1, P(q), P(k), P(vt), P(ctx), B, H, nv, nv, st)
        for _ in range(2):
            assert f() == 0, lib.rz_last_error()
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        per[nv] = ms * 1e3 / (B * H * (nv // 256) / 256.0)
        print(f"{v:4d} {names.get(v, ''):20s} nv {nv:5d} {ms:7.3f} ms  {4.0 * B * H * nv * nv * 64 / ms / 1e9:7.1f} TFLOP/s-equivalent   {per[nv]:7.2f} us per workgroup-slot ({nv // 64} tiles)", flush=True)
    if len(nvs) >= 2:
        a, b = nvs[-2], nvs[-1]
        slope = (per[b] - per[a]) / ((b - a) // 64)
        print(f"     -> {slope * 1e3:7.1f} ns per tile, {per[b] - slope * (b // 64):6.2f} us per workgroup outside the loop", flush=True)
