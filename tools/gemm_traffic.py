"""Launch the headline shapes' GEMMs through rz_gemm_ex with a forced kernel variant / tile walk (to be run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE`; tools/gemm_traffic_summary.py reads the counter files).
  python3 tools/gemm_traffic.py --variant 12 --raster 8      shapes: q|k|v-like N=2304 (EPI_STORE), fc1 N=3072 (EPI_GELU), fc2 K=3072 (fp32 RMW)"""
import argparse, ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=8)
ap.add_argument("--raster", type=int, default=0)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 32 * 5376
lib.rz_set_option(b"gemm_variant", a.variant)
lib.rz_set_option(b"gemm_raster", a.raster)
g = torch.Generator(device="cuda").manual_seed(1)
for (N, K, epi) in ((2304, 768, 0), (3072, 768, 1), (768, 3072, 4), (768, 768, 4)):
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.7).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    scale = torch.rand(N, device="cuda", generator=g)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    resid = torch.zeros(M, N, device="cuda") if epi == 4 else None
    for _ in range(a.reps):
        rc = lib.rz_gemm_ex(1, epi, P(x), K, P(w), K, P(bias), P(out), N, P(scale), P(resid) if resid is not None else None, N, 5376, N // 64, M, N, K, st)
        assert rc == 0, lib.rz_last_error()
    torch.cuda.synchronize()
    del x, w, out, resid
print("done", a.variant, a.raster)
