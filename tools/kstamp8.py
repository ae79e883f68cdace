"""[tools build only: run with RZ_EXPERIMENTS=1 in the environment — the stamped kernels are not in the production library]
Diagnostic (GPU box): where the persistent GEMM (gemm8.hip, stamped build) spends its time — per wave the 100 MHz ticks
inside K loops and inside epilogues, and how synchronised the epilogue starts of different CUs are.
  python tools/kstamp8.py [epi=2|1|4] [N] [K] [images]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 768
images = int(sys.argv[4]) if len(sys.argv) > 4 else 32
skew = int(sys.argv[5]) if len(sys.argv) > 5 else 0
lda_mul = int(sys.argv[6]) if len(sys.argv) > 6 else 1      # 0: every A row is row 0 (A always L2-resident): isolates the cost of streaming A
ldw_mul = int(sys.argv[7]) if len(sys.argv) > 7 else 1
M = images * 5376
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
bias = torch.randn(N, device="cuda")
scale = torch.rand(N, device="cuda")
resid = torch.randn(M, N, device="cuda") if epi == 4 else None
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(256 * 8 * 32, device="cuda", dtype=torch.int64)
lib.rz_set_option(b"gemm_variant", 8)
if skew:
    pass  # the start-up skew experiment of round 2 is gone (profiles/r02/gemm_v8_startup_skew_sweep.log)
f = lambda: lib.rz_gemm_ex(1, epi, P(a), K * lda_mul, P(w), K * ldw_mul, P(bias), P(out), N, P(scale), P(resid), N, 5376, N // 64, M, N, K, ST())
for _ in range(3):
    assert f() == 0, lib.rz_last_error()
torch.cuda.synchronize()
assert lib.rz_debug_buffer(b"gemm_v8_stamps", P(dbg)) == 0
assert f() == 0, lib.rz_last_error()
torch.cuda.synchronize()
lib.rz_debug_buffer(b"gemm_v8_stamps", None)
d = dbg.cpu().view(256, 8, 32)
n = d[:, :, 2].float()
ok = n[:, 0] > 0
k_us = (d[:, :, 0].float() / n.clamp(min=1) * 0.01)[ok]
e_us = (d[:, :, 1].float() / n.clamp(min=1) * 0.01)[ok]
life = ((d[:, :, 4] - d[:, :, 3]).float() * 0.01)[ok]
print(f"epi {epi} M={M} N={N} K={K} skew={skew} lda*{lda_mul} ldw*{ldw_mul}: {int(ok.sum())} workgroups, tiles per WG {n[ok][:,0].min():.0f}-{n[ok][:,0].max():.0f}")
print(f"K loop per tile  : group0 waves {k_us[:, :4].mean():.2f} us, group1 waves {k_us[:, 4:].mean():.2f} us   ({K // 64} K tiles -> {k_us.mean() / (K // 64):.3f} us per K tile)")
clk = (d[:, :, 5].float() / d[:, :, 0].float().clamp(min=1) * 0.1)[ok]
print(f"shader clock inside the K loops: mean {clk.mean():.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f}) -> {k_us.mean() / (K // 64) * clk.mean() * 1e3:.0f} cycles per K tile (pure MFMA: 2048)")
print(f"epilogue per tile: group0 waves {e_us[:, :4].mean():.2f} us (min {e_us[:, :4].min():.2f} max {e_us[:, :4].max():.2f}), group1 waves {e_us[:, 4:].mean():.2f} us")
print(f"workgroup lifetime: mean {life[:, 0].mean():.1f} us, min {life[:, 0].min():.1f}, max {life[:, 0].max():.1f}")
t0 = d[:, 0, 3][ok].min()
starts = ((d[:, 0, 8:32][ok] - t0).float() * 0.01)
for t in range(min(8, int(n[ok][:, 0].min()))):
    col = starts[:, t]
    print(f"epilogue #{t} start over workgroups: min {col.min():7.1f} us  median {col.median():7.1f}  max {col.max():7.1f}  (spread {col.max() - col.min():.1f})")
