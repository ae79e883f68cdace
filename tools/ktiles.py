"""Where do 256 x 256 tiles (persistent gemm8) beat 128 x 128 ones for the N = 768 GEMMs (out-proj K = 768, fc2 K = 3072) and the wide ones (q|k N = 1536,
fc1 N = 3072) at SMALL row counts?  One process, interleaved rounds, rz_gemm_ex with gemm_variant forced to 1 / 8, M = row_tiles x 256.  Calibrates
gemm.hip::big_tiles_pay.   python tools/ktiles.py"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib  # noqa: E402
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tdt = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, epi, N, K in (("out-proj", 4, 768, 768), ("fc2", 4, 768, 3072), ("q|k", 2, 1536, 768), ("fc1", 1, 3072, 768)):
    print(f"== {name}: N = {N}, K = {K}   (us per launch: 128x128 | 256x256 persistent; tiles256 = row_tiles x {N // 256})")
    for rt in (11, 21, 22, 33, 42, 44, 47, 55, 63, 66, 84, 88, 105, 110, 132, 176):
        M = rt * 256
        npad = 256
        a = torch.randn(M, K, device="cuda").to(tdt)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
        bias = torch.randn(N, device="cuda")
        scale = torch.rand(N, device="cuda")
        resid = torch.randn(M, N, device="cuda") if epi == 4 else None
        out = torch.empty(M, N, device="cuda", dtype=tdt)
        f = lambda: lib.rz_gemm_ex(1, epi, P(a), K, P(w), K, P(bias), P(out), N, P(scale), P(resid), N, npad, N // 64, M, N, K, ST())
        t = {1: [], 8: []}
        for _ in range(3):
            for v in (1, 8):
                lib.rz_set_option(b"gemm_variant", v)
                assert f() == 0, lib.rz_last_error()
                t[v].append(timeit(f))
        lib.rz_set_option(b"gemm_variant", 0)
        m1, m8 = sorted(t[1])[1], sorted(t[8])[1]
        t256 = rt * (N // 256)
        print(f"row tiles {rt:4d}  tiles256 {t256:5d}  fullness {t256 / (((t256 + 255) // 256) * 256):.2f}:  {m1:7.1f} | {m8:7.1f}   -> {'256' if m8 < m1 else '128'} ({m1 / m8:.2f})", flush=True)
