"""GPU box: the per-request leg of bench.py (one image + one NEW text per request) A/B — fp32 mode with the predicated overflow guard on / off
(what the ~75 empty launches of the guard's second pass cost per request), bf16 for reference.   python tools/request_ab.py [side]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from radzero_amd import _lib  # noqa: E402
from radzero_amd.config import RadZeroConfig  # noqa: E402
from radzero_amd.weights import make_state_dict  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = RadZeroConfig()
sd = make_state_dict(cfg, 20260103)
dev = torch.device("cuda", 0)
lib = _lib.load()
keys = ("ms_per_request", "ms_per_request_text_embedding_supplied", "text_encoder_share_ms", "text_encoder_alone_ms", "ms_per_request_hipgraph")
mx_ab = len(sys.argv) > 2 and sys.argv[2] == "mx"      # second mode: the MX operand form forced at batch 1 (gemm_f32_mx = 2) against the default (three planes below 64 row tiles)
legs = ((("f32 default form", "f32", 1, 1), ("f32 MX form forced (gemm_f32_mx = 2)", "f32", 1, 2), ("f32 default form (again)", "f32", 1, 1), ("f32 MX forced (again)", "f32", 1, 2)) if mx_ab else
        (("bf16", "bf16", 1, 1), ("f32 guard on", "f32", 1, 1), ("f32 guard off", "f32", 0, 1), ("f32 guard on (again)", "f32", 1, 1), ("bf16 (again)", "bf16", 1, 1)))
for label, dtype, guard, mx in legs:
    _lib.check(lib.rz_set_option(b"f32_split_guard", guard), "rz_set_option")
    _lib.check(lib.rz_set_option(b"gemm_f32_mx", mx), "rz_set_option")
    r = bench.request_leg(sd, cfg, dev, dtype=dtype, S=side, steps=16, warmup=4)
    print(json.dumps({"leg": label, "side": side, **{k: r[k] for k in keys}}), flush=True)
