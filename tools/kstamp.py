"""[tools build only: run with RZ_EXPERIMENTS=1 in the environment — the stamped kernels are not in the production library]
Diagnostic (GPU box): in-kernel s_memtime stamps of one K tile of the staggered 256x256 GEMM (gemm7.hip MODE 2).
  python tools/kstamp.py [N] [K]"""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib
lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
M = 43008
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(8 * 24 + 8, device="cuda", dtype=torch.int64)
lib.rz_set_option(b"gemm_variant", 9)
for _ in range(5):
    rc = lib.rz_gemm_ex(1, 0, P(a), K, P(w), K, P(bias), P(out), N, None, P(dbg), 0, M, N // 64, M, N, K, ST())
    assert rc == 0, lib.rz_last_error()
torch.cuda.synchronize()
raw = dbg.cpu()
st = raw[:192].view(8, 24)
print(f'main loop of WG0 (steady-state K tiles 0..nk-3, stamped): {raw[192].item()} shader cycles, {raw[193].item()} ticks of 100 MHz -> clock {raw[192].item() / max(raw[193].item(), 1) * 0.1:.2f} GHz, {raw[192].item() / (K // 64 - 2):.0f} cycles per K tile')
t0 = st[:, 0].min().item()
names = ["LOAD.start", "LOAD.issued", "bar1.passed", "MFMA.issued", "vmcnt.done", "bar2.passed"]
print("stamps relative to the earliest wave's tile start (shader cycles); one row per wave (0-3 group 0, 4-7 group 1)")
for wv in range(8):
    row = []
    for u in range(4):
        row.append(" ".join(f"{st[wv, u * 6 + i].item() - t0:5d}" for i in range(6)))
    print(f"w{wv}: " + " | ".join(row))
print("per-wave deltas, phase by phase: [LOAD issue, wait@bar1, MFMA issue, vmcnt wait, wait@bar2]")
for wv in range(8):
    row = []
    for u in range(4):
        d = [st[wv, u * 6 + i + 1].item() - st[wv, u * 6 + i].item() for i in range(5)]
        row.append(" ".join(f"{x:4d}" for x in d))
    tot = st[wv, 23].item() - st[wv, 0].item()
    print(f"w{wv}: " + " | ".join(row) + f" | tile {tot}")
