"""Time rz_upsample_maps_ex alone (cfg 4's map set: 1024 maps of 73 x 73 -> 1024 x 1024 fp32 = 4.29 GB per launch)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd import _lib
lib = _lib.load(auto_build=False)
M, g, S = 1024, 73, 1024
maps = torch.randn(M, g * g, device="cuda")
out = torch.empty(M, S, S, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f = lambda sg: lib.rz_upsample_maps_ex(None, P(maps), g * g, M, g, S, S, sg, 0, P(out), st)
for sg in (0, 1):
    for _ in range(3): assert f(sg) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f(sg)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{os.environ.get('RZ_LIB_PATH', 'default'):40s} sigmoid={sg}: {ms:.3f} ms  {M * S * S * 4 / ms / 1e9:.2f} TB/s", flush=True)
