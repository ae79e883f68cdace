"""Diagnostic (GPU box): MLP pair fc1(+GELU) -> fc2(+LayerScale residual) over B images, whole batch at once vs in row
chunks that reuse ONE hidden buffer small enough to stay in the Infinity Cache.   python tools/kmlp.py [images]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
images = int(sys.argv[1]) if len(sys.argv) > 1 else 32
npad = 5376
M = images * npad
D, H = 768, 3072
xn = torch.randn(M, D, device="cuda").bfloat16()
w1 = (torch.randn(H, D, device="cuda") / math.sqrt(D)).bfloat16()
w2 = (torch.randn(D, H, device="cuda") / math.sqrt(H)).bfloat16()
b1 = torch.randn(H, device="cuda"); b2 = torch.randn(D, device="cuda"); ls = torch.rand(D, device="cuda")
h = torch.randn(M, D, device="cuda")


def timeit(fn, iters=5, warm=2):
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


def ptr(t, row, width, esz):
    return ctypes.c_void_p(t.data_ptr() + row * width * esz)


for chunk_images in (images, 16, 8, 4, 2):
    if chunk_images > images:
        continue
    mc = chunk_images * npad
    hid = torch.empty(mc, H, device="cuda", dtype=torch.bfloat16)

    def run():
        for r0 in range(0, M, mc):
            rc = lib.rz_gemm_ex(1, 1, ptr(xn, r0, D, 2), D, P(w1), D, P(b1), P(hid), H, None, None, 0, npad, H // 64, mc, H, D, ST())
            assert rc == 0, lib.rz_last_error()
            rc = lib.rz_gemm_ex(1, 4, P(hid), H, P(w2), H, P(b2), None, 0, P(ls), ptr(h, r0, D, 4), D, npad, D // 64, mc, D, H, ST())
            assert rc == 0, lib.rz_last_error()
    ms = timeit(run)
    fl = 2 * 2.0 * M * D * H
    print(f"MLP pair {images} images in chunks of {chunk_images} (hidden buffer {mc * H * 2 / 2**20:.0f} MiB): {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")
