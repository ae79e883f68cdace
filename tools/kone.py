"""Run ONE kernel configuration a few times (for rocprofv3 --pmc passes).  python tools/kone.py gemm|attn [epi N K]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
what = sys.argv[1]
images = int(os.environ.get("IMAGES", "8"))
n = 5330; npad = 5376
if os.environ.get("V1"):
    lib.rz_set_option(b"gemm_v1_only", 1)
if what == "gemm":
    epi, N, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    M = images * npad
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device="cuda"); scale = torch.rand(N, device="cuda")
    resid = torch.randn(M, N, device="cuda") if epi == 4 else None
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        assert lib.rz_gemm_ex(1, epi, P(a), K, P(w), K, P(bias), P(out), N, P(scale), P(resid), N, npad, N // 64, M, N, K, ST()) == 0
else:
    H = 12
    q = (torch.randn(images, H, npad, 64, device="cuda") * 0.6).bfloat16()
    k = torch.randn(images, H, npad, 64, device="cuda").bfloat16()
    vt = torch.randn(images, H, 64, npad, device="cuda").bfloat16()
    ctx = torch.empty(images * npad, H * 64, device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        assert lib.rz_flash_attention(1, P(q), P(k), P(vt), P(ctx), images, H, n, npad, ST()) == 0
torch.cuda.synchronize()
