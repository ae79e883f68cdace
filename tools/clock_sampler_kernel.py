"""Sustained shader clock / socket power of ONE attention variant run back to back for a few seconds (RZ_EXPERIMENTS=1 for variants
beyond the product's): python3 tools/clock_sampler_kernel.py <attn_variant> ...  — rocm-smi polled from a thread while the launches run."""
import ctypes, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
lib = _lib.load()
B, H, nv, npad = 32, 12, 5330, 5376
P = lambda t: ctypes.c_void_p(t.data_ptr())
q = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
k = (torch.randn(B, H, npad, 64, device="cuda") * 0.5).bfloat16()
vt = torch.randn(B, H, 64, npad, device="cuda").bfloat16()
ctx = torch.empty(B, npad, H * 64, device="cuda", dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
stop, samples = False, []
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        c = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out); p = re.findall(r"Power \(W\): ([\d.]+)", out)
        if c and p: samples.append((int(c[0]), float(p[0])))
        time.sleep(0.2)
for arg in sys.argv[1:]:
    v = int(arg)
    lib.rz_set_option(b"attn_variant", v)
    f = lambda: lib.rz_flash_attention(1, P(q), P(k), P(vt), P(ctx), B, H, nv, npad, st)
    for _ in range(5): assert f() == 0
    torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=poll); th.start()
    n = 2500
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    stop = True; th.join()
    s = samples[2:-1] or samples
    print(f"variant {v:4d}: {ms:.3f} ms per launch sustained over {n} launches; sclk {sum(c for c, _ in s) / len(s):.0f} MHz, power {sum(p for _, p in s) / len(s):.0f} W ({len(s)} samples)", flush=True)
