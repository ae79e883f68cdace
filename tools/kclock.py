"""GEMM timing-ablation probe (tools/gen_gemm10_kloop.py RZ_V10_ABLATE=...): run ONE fc2-shaped GEMM (M = 32 x 5376, N = 768, K = 3072, bf16,
EPI_RESID_SCALE) repeatedly with the kernel variant given on the command line, for rocprofv3 --pmc to attribute clock and MFMA-busy to it.
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -- python3 tools/kclock.py 10
Results of an ablated library are WRONG by construction; only the timing matters."""
import ctypes, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
lib = _lib.load()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shape = sys.argv[2] if len(sys.argv) > 2 else "fc2"
N, K, epi = {"fc2": (768, 3072, 4), "fc1": (3072, 768, 1), "qk": (1536, 768, 2)}[shape]
M = 32 * 5376
P = lambda t: ctypes.c_void_p(t.data_ptr())
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
bias, scale = torch.randn(N, device="cuda"), torch.rand(N, device="cuda")
resid = torch.randn(M, N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
lib.rz_set_option(b"gemm_variant", variant)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda: lib.rz_gemm_ex(1, epi, P(a), K, P(w), K, P(bias), P(out), N, P(scale), P(resid), N, 5376, N // 64, M, N, K, st)
for _ in range(5):
    assert f() == 0, lib.rz_last_error()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 40
for _ in range(n):
    f()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"variant {variant} {shape}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
