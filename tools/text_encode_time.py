"""GPU box: prompt-encode time (GPU events and wall) per dtype and split option, for 1 / 14 / 64 prompts.   python tools/text_encode_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.modeling import RadZeroModel  # noqa: E402
from radzero_amd.synthetic import synthetic_prompts  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

cfg = RadZeroConfig()
sd = make_state_dict(cfg, 20260103)
for dtype, opts in ((torch.bfloat16, {}), (torch.float32, {}), (torch.float32, {"f32_split_guard": 0}), (torch.float32, {"gemm_f32_split": 0})):
    m = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=dtype, device="cuda:0").eval()
    m.text_cache_enabled = False
    for k, v in opts.items():
        m.set_model_option(k, v)
    for T, lo, hi in ((1, 12, 12), (14, 6, 10), (64, 8, 32)):
        ids, mask = synthetic_prompts(T, lo, hi, 5)
        enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
        for _ in range(3):
            m.forward_text_model(enc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            m.forward_text_model(enc)
        e1.record()
        host = (time.perf_counter() - t0) / reps * 1e3
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        print(f"{str(dtype):15s} {opts!s:28s} T={T:3d}: GPU {e0.elapsed_time(e1) / reps:6.3f} ms per encode, host launch time {host:6.3f} ms, wall {wall:6.3f} ms", flush=True)
    m.close()
