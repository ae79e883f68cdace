"""Write the synthetic checkpoint of a reduced-depth model (real widths: 768 / 3072, so every packing path runs at its real leading
dimensions) as a flat binary for tools/host_asan/driver.cpp:  u32 n; n x { u32 name_len, name, u64 numel, float32 data }."""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from radzero_amd.config import RadZeroConfig
from radzero_amd.weights import make_state_dict

out = sys.argv[1]
cfg = RadZeroConfig(vit_layers=2, align_layers=1, text_layers=1, vocab_size=512)
sd = make_state_dict(cfg, 5)
with open(out, "wb") as f:
    f.write(struct.pack("<I", len(sd)))
    for name, v in sd.items():
        a = np.ascontiguousarray(np.asarray(v, np.float32)).reshape(-1)
        nb = name.encode()
        f.write(struct.pack("<I", len(nb))); f.write(nb); f.write(struct.pack("<Q", a.size)); f.write(a.tobytes())
print(f"{len(sd)} tensors -> {out}")
