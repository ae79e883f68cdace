"""Diagnostic (GPU box): per-K-tile cost of a GEMM variant = slope of time over K at fixed M, N (epilogue on / skipped).
  python tools/kslope.py [variant] [N]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 7
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
M = int(os.environ.get('KSLOPE_M', 43008))
tiles = (M // 256) * (N // 256)
rounds = tiles / 256.0


def timeit(fn, iters=20, warm=5):
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


lib.rz_set_option(b"gemm_variant", variant)
for flags in [int(x) for x in os.environ.get('KSLOPE_FLAGS', '0,4').split(',')]:
    lib.rz_set_option(b"gemm_debug_flags", flags)
    res = []
    for K in (256, 768, 1536, 3072):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, device="cuda")
        if os.environ.get("KSLOPE_ZEROS"):      # DVFS probe: same instruction stream, no operand toggling
            a.zero_(); w.zero_()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: lib.rz_gemm_ex(1, 0, P(a), K, P(w), K, P(bias), P(out), N, None, None, 0, M, N // 64, M, N, K, ST())
        assert f() == 0, lib.rz_last_error()
        ms = timeit(f)
        res.append((K, ms))
        print(f"variant {variant} flags {flags} N={N} K={K}: {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s  {ms * 1e3 / rounds:.2f} us per tile-round")
    (k0, t0), (k1, t1) = res[1], res[3]
    b = (t1 - t0) * 1e3 / rounds / ((k1 - k0) / 64)
    print(f"  -> {b:.3f} us per K tile (slope 768..3072), fixed cost {t0 * 1e3 / rounds - b * k0 / 64:.2f} us per tile")
