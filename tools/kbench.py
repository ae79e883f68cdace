"""Micro-benchmarks of the two dominant kernels at the bench shapes (GPU box only).
  python tools/kbench.py attn|gemm|all [--images 8]"""
import argparse
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bench_attn(images, n=5330, dt=1, zeros=False):
    npad = (n + 127) // 128 * 128
    H = 12
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    q = (torch.randn(images, H, npad, 64, device="cuda") * 0.6).to(tdt)
    k = torch.randn(images, H, npad, 64, device="cuda").to(tdt)
    vt = torch.randn(images, H, 64, npad, device="cuda").to(tdt)
    if zeros:      # DVFS probe: same instruction stream, no operand toggling
        q.zero_(); k.zero_(); vt.zero_()
    ctx = torch.empty(images * npad, H * 64, device="cuda", dtype=tdt)
    f = lambda: lib.rz_flash_attention(dt, P(q), P(k), P(vt), P(ctx), images, H, n, npad, ST())
    assert f() == 0, lib.rz_last_error()
    ms = timeit(f)
    fl = images * 4.0 * n * n * 768
    print(f"attn dt={dt} images={images} n={n}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s (algorithmic)")


def bench_gemm(images, n=5330, dt=1):
    npad = (n + 127) // 128 * 128
    M = images * npad
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    shapes = [("qk    EPI_HEADS", 2, 1536, 768), ("v     EPI_VT", 3, 768, 768), ("out   EPI_RESID_SCALE", 4, 768, 768),
              ("fc1   EPI_GELU", 1, 3072, 768), ("fc2   EPI_RESID_SCALE", 4, 768, 3072), ("plain EPI_STORE", 0, 3072, 768)]
    tot = 0.0
    for name, epi, N, K in shapes:
        a = torch.randn(M, K, device="cuda").to(tdt)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
        bias = torch.randn(N, device="cuda")
        scale = torch.rand(N, device="cuda")
        resid = torch.randn(M, N, device="cuda") if epi == 4 else None
        out = torch.empty(M, N, device="cuda", dtype=tdt)
        heads = N // 64
        f = lambda: lib.rz_gemm_ex(dt, epi, P(a), K, P(w), K, P(bias), P(out), N, P(scale), P(resid), N, npad, heads, M, N, K, ST())
        assert f() == 0, lib.rz_last_error()
        ms = timeit(f)
        if "plain" not in name:
            tot += ms
        print(f"gemm {name:24s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")
    print(f"gemm per-layer total {tot:.3f} ms for {images} images")


def bench_gemm_ab(images, n=5330, dt=1, variants=(7, 8), rounds=3):
    """A/B of GEMM kernels in ONE process, interleaved rounds (same tensors, same clocks): per shape the median ms of
    each variant, plus the block's q|k|v projection as one fused launch (variant 8) against the two separate ones."""
    npad = (n + 127) // 128 * 128
    M = images * npad
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    shapes = [("qk    EPI_HEADS", 2, 1536, 768), ("v     EPI_VT", 3, 768, 768), ("out   EPI_RESID_SCALE", 4, 768, 768),
              ("fc1   EPI_GELU", 1, 3072, 768), ("fc2   EPI_RESID_SCALE", 4, 768, 3072), ("patch EPI_PATCH", 6, 768, 640)]
    tot = {v: 0.0 for v in variants}
    for name, epi, N, K in shapes:
        a = torch.randn(M, K, device="cuda").to(tdt)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
        bias = torch.randn(N, device="cuda")
        scale = torch.rand(npad, N, device="cuda") if epi == 6 else torch.rand(N, device="cuda")
        resid = torch.randn(M, N, device="cuda") if epi == 4 else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == 6 else tdt)
        f = lambda: lib.rz_gemm_ex(dt, epi, P(a), K, P(w), K, None if epi == 6 else P(bias), P(out), N, P(scale), P(resid), N, npad, N // 64, M, N, K, ST())
        ms = {v: [] for v in variants}
        for _ in range(rounds):
            for v in variants:
                lib.rz_set_option(b"gemm_variant", v)
                assert f() == 0, lib.rz_last_error()
                ms[v].append(timeit(f, iters=8, warm=2))
        lib.rz_set_option(b"gemm_variant", 0)
        line = f"gemm {name:24s} M={M} N={N} K={K}:"
        for v in variants:
            med = sorted(ms[v])[len(ms[v]) // 2]
            if "patch" not in name:
                tot[v] += med
            line += f"  v{v} {med:.3f} ms {2.0 * M * N * K / med / 1e9:7.1f} TF"
        print(line, flush=True)
    print("gemm per-layer total (qk + v + out + fc1 + fc2): " + "  ".join(f"v{v} {tot[v]:.3f} ms" for v in variants) + f"  for {images} images")
    x = torch.randn(M, 768, device="cuda").to(tdt)
    w = (torch.randn(2304, 768, device="cuda") / math.sqrt(768)).to(tdt)
    bias = torch.randn(2304, device="cuda")
    qk = torch.empty(images, 24, npad, 64, device="cuda", dtype=tdt)
    vt = torch.empty(images, 12, 64, npad, device="cuda", dtype=tdt)
    fused = ctypes.c_int(0)
    f = lambda: lib.rz_gemm_qkv(dt, P(x), P(w), P(bias), P(qk), P(vt), npad, 12, M, ctypes.byref(fused), ST())
    res = {}
    for _ in range(rounds):
        for v in (0,) + tuple(variants):
            lib.rz_set_option(b"gemm_variant", v)
            assert f() == 0, lib.rz_last_error()
            res.setdefault((v, fused.value), []).append(timeit(f, iters=8, warm=2))
    lib.rz_set_option(b"gemm_variant", 0)
    for (v, fu), t in res.items():
        med = sorted(t)[len(t) // 2]
        print(f"qkv projection variant {v} ({'ONE fused launch' if fu else 'q|k + v launches'}): {med:.3f} ms {2.0 * M * 2304 * 768 / med / 1e9:7.1f} TF")


def bench_library(images, n=5330, dt=1):
    """Calibration only (never on the product path): the vendor libraries at the same shapes -- hipBLASLt through
    F.linear (bias fused, no other epilogue) and torch SDPA (its flash backend) on the padded token count."""
    import torch.nn.functional as F
    npad = (n + 127) // 128 * 128
    M = images * npad
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    for name, N, K in [("qk", 1536, 768), ("v/out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
        a = torch.randn(M, K, device="cuda").to(tdt)
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).to(tdt)
        bias = torch.randn(N, device="cuda").to(tdt)
        ms = timeit(lambda: F.linear(a, w, bias))
        ms0 = timeit(lambda: F.linear(a, w))
        print(f"hipblaslt {name:6s} M={M} N={N} K={K}: +bias {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s | "
              f"no bias {ms0:.3f} ms {2.0 * M * N * K / ms0 / 1e9:.1f} TFLOP/s")
    q = (torch.randn(images, 12, n, 64, device="cuda") * 0.6).to(tdt)
    k = torch.randn(images, 12, n, 64, device="cuda").to(tdt)
    v = torch.randn(images, 12, n, 64, device="cuda").to(tdt)
    fl = images * 4.0 * n * n * 768
    for backend in ("FLASH_ATTENTION", "EFFICIENT_ATTENTION"):
        try:
            from torch.nn.attention import SDPBackend, sdpa_kernel
            with sdpa_kernel(getattr(SDPBackend, backend)):
                ms = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
            print(f"torch sdpa {backend} images={images} n={n}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
        except Exception as e:  # backend not built into this wheel
            print(f"torch sdpa {backend}: unavailable ({type(e).__name__}: {str(e)[:120]})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--images", type=int, default=8)
    ap.add_argument("--dtype", type=int, default=1)
    ap.add_argument("--v1", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--attn-variant", type=int, default=0)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--tokens", type=int, default=5330, help="tokens per image for the attention bench (5330 = 1024^2, 11882 = 1536^2)")
    ap.add_argument("--variants", type=int, nargs="+", default=[8, 10], help="GEMM kernels of the `gemmab` A/B (interleaved rounds in one process)")
    a = ap.parse_args()
    if a.v1:
        lib.rz_set_option(b"gemm_v1_only", 1)
    if a.attn_variant:
        lib.rz_set_option(b"attn_variant", a.attn_variant)
    if a.variant:
        lib.rz_set_option(b"gemm_variant", a.variant)
    if a.what in ("attn", "all"):
        bench_attn(a.images, n=a.tokens, dt=a.dtype, zeros=a.zeros)
    if a.what in ("gemm", "all"):
        bench_gemm(a.images, n=a.tokens, dt=a.dtype)
    if a.what == "gemmab":
        bench_gemm_ab(a.images, dt=a.dtype, variants=tuple(a.variants))
    if a.what == "library":
        bench_library(a.images, dt=a.dtype)
