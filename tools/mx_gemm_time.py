"""Time rz_gemm_f32_split (fp32 GEMM forms, RMW epilogue, gemm7's loop for both) over K: the slope is the cost of a K tile of each kind."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
lib = _lib.load(auto_build=False)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N = 32 * 5376, 768
for K in (256, 768, 1536, 3072):
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / math.sqrt(K)
    bias = torch.randn(N, device="cuda"); ones = torch.ones(N, device="cuda"); out = torch.zeros(M, N, device="cuda")
    ws_a = torch.empty(M * K * 6, dtype=torch.uint8, device="cuda"); ws_w = torch.empty(N * K * 6, dtype=torch.uint8, device="cuda")
    for form in (0, 1):
        f = lambda: lib.rz_gemm_f32_split(form, P(a), P(w), P(bias), P(ones), P(out), P(ws_a), P(ws_w), M, N, K, st)
        for _ in range(2): assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        print(f"K={K:5d} form {form}: {e0.elapsed_time(e1) / 5:.3f} ms per call (incl. the two split kernels)", flush=True)
    del a, w, out, ws_a, ws_w
