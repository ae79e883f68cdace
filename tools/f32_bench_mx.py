"""fp32-mode forward of the headline shape with the MX form on / off (env RZ_MX), for rocprofv3 (tools/f32_mx_profile.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radzero_amd.config import RadZeroConfig
from radzero_amd.modeling import RadZeroModel
from radzero_amd.weights import make_state_dict
cfg = RadZeroConfig()
m = RadZeroModel.from_state_dict(make_state_dict(cfg, 20260103), cfg, torch_dtype=torch.float32, device="cuda:0").eval()
m.set_model_option("gemm_f32_mx", int(os.environ.get("RZ_MX", "1")))
px = torch.randn(32, 3, 1024, 1024, device="cuda")
for _ in range(3):
    m.forward_vision_model(px)
torch.cuda.synchronize()
print("done")
