import ctypes, math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from radzero_amd import _lib
lib = _lib.load(auto_build=False)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, outl) in ((512, 768, 768, 40.0), (512, 768, 768, 1.0), (1024, 3072, 768, 1.0), (768, 768, 3072, 1.0), (512, 768, 640, 1.0)):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-2, 2, (M, 1), generator=g).float())
    a[:, 5] *= outl
    w = torch.randn(N, K, generator=g) / math.sqrt(K) * torch.exp2(torch.randint(-1, 2, (N, 1), generator=g).float())
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    mag = a.double().abs() @ w.double().abs().t()
    rms = torch.sqrt((a.double() ** 2) @ (w.double() ** 2).t())
    ad, wd, bd, ones = a.cuda(), w.cuda(), bias.cuda(), torch.ones(N, device="cuda")
    for form in (0, 1):
        out = torch.zeros(M, N, device="cuda")
        ws_a = torch.empty(M * K * 6, dtype=torch.uint8, device="cuda"); ws_w = torch.empty(N * K * 6, dtype=torch.uint8, device="cuda")
        rc = lib.rz_gemm_f32_split(form, P(ad), P(wd), P(bd), P(ones), P(out), P(ws_a), P(ws_w), M, N, K, st)
        assert rc == 0, lib.rz_last_error()
        torch.cuda.synchronize()
        e = (out.double().cpu() - ref).abs()
        print(f"M{M} N{N} K{K} outlier x{outl:g} form {form}: max|err| {float(e.max()):.3e}  max err/sum|ab| {float((e / mag).max()):.3e}  max err/rms {float((e / rms).max()):.3e}  rms err/rms {float(torch.sqrt((e**2).mean()) / rms.mean()):.3e}")
# fp32 torch matmul for comparison
    o32 = (ad @ wd.t() + bd).double().cpu()
    e = (o32 - ref).abs()
    print(f"      torch fp32 matmul: max err/rms {float((e / rms).max()):.3e}")
