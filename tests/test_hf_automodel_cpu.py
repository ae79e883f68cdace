"""The AutoModel surface without a GPU (README.md:72-89): a directory in the reference's save_pretrained layout,
after radzero_amd.hf.export_auto_map, dispatches `AutoConfig` / `AutoModel.from_pretrained(dir, trust_remote_code=True, ...)` to the HIP
classes in a FRESH interpreter that never imported radzero_amd itself — the call then stops where the product must stop on a box
without a HIP device: loudly, with no CPU fallback."""
import json
import os
import shutil
import subprocess
import sys

from conftest import GOLDEN_DIR, ROOT

CHILD = r"""
import sys, torch
from transformers import AutoConfig, AutoModel
d = sys.argv[1]
assert "radzero_amd" not in sys.modules
c = AutoConfig.from_pretrained(d, trust_remote_code=True)
print("CONFIG", type(c).__name__, c.model_type, sorted(c.auto_map))
try:
    AutoModel.from_pretrained(d, trust_remote_code=True, torch_dtype=torch.float32, device_map=torch.device("cuda"))
    print("MODEL built")
except RuntimeError as e:
    print("MODEL RuntimeError:", e)
"""


def _layout(tmp_path):
    from radzero_amd.checkpoint import save_checkpoint
    from radzero_amd.config import RadZeroConfig
    from radzero_amd.weights import make_state_dict
    keys = json.load(open(os.path.join(GOLDEN_DIR, "hf_layout", "keys.json")))
    cfg = RadZeroConfig(**keys["radzero_config"])
    src = tmp_path / "ref"
    src.mkdir()
    shutil.copy(os.path.join(GOLDEN_DIR, "hf_layout", "config.json"), src / "config.json")
    save_checkpoint(make_state_dict(cfg, keys["weights_seed"]), str(src))
    return cfg, src


def test_export_auto_map_and_dispatch_in_a_fresh_interpreter(tmp_path):
    from radzero_amd.hf import AUTO_MAP, MODEL_TYPE, export_auto_map
    cfg, src = _layout(tmp_path)
    before = open(src / "config.json").read()
    dst = export_auto_map(str(src), str(tmp_path / "local"))
    assert open(src / "config.json").read() == before
    written = json.load(open(os.path.join(dst, "config.json")))
    assert written["auto_map"] == AUTO_MAP and written["model_type"] == MODEL_TYPE
    assert written["vision_config"] == json.loads(before)["vision_config"]           # everything else as the reference wrote it
    assert os.path.islink(os.path.join(dst, "model.safetensors"))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HF_HOME=str(tmp_path / "hf"),
               HF_MODULES_CACHE=str(tmp_path / "hf" / "modules"), HF_HUB_OFFLINE="1")
    r = subprocess.run([sys.executable, "-c", CHILD, dst], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "CONFIG RadZeroHFConfig radzero_hip ['AutoConfig', 'AutoModel']" in r.stdout, r.stdout
    import torch
    if not torch.cuda.is_available():
        assert "MODEL RuntimeError: no HIP device visible" in r.stdout, r.stdout     # reached RadZeroModel.__init__: no fallback


def test_hf_config_gives_the_kernels_config(tmp_path):
    from radzero_amd.checkpoint import load_checkpoint
    from radzero_amd.hf import RadZeroHFConfig, export_auto_map
    from transformers import AutoConfig
    cfg, src = _layout(tmp_path)
    import pytest
    with pytest.raises(ValueError, match="out_dir"):           # mutating the source checkpoint must be asked for explicitly
        export_auto_map(str(src))
    before = open(src / "config.json").read()
    assert export_auto_map(str(src), str(tmp_path / "exported")) != str(src) and open(src / "config.json").read() == before      # out_dir: the original is untouched
    export_auto_map(str(src), in_place=True)
    c = AutoConfig.from_pretrained(str(src))                   # model_type is registered by importing radzero_amd.hf: no remote code needed
    assert isinstance(c, RadZeroHFConfig)
    assert c.to_radzero(load_checkpoint(str(src))) == cfg


def test_save_pretrained_round_trip(tmp_path):
    """`model.save_pretrained(dir)` (radzero_amd.hf): weights under the reference's names + config.json in the reference's layout + auto_map;
    the directory reads back to the same RadZeroConfig and tensors, for the released head and for a projected global_alignment one.  Host
    logic only: the model object is a stand-in carrying what save_pretrained reads."""
    import numpy as np
    import torch
    from transformers import AutoConfig
    from radzero_amd.checkpoint import config_from_hf, load_checkpoint
    from radzero_amd.config import RadZeroConfig
    from radzero_amd.hf import RadZeroHFConfig, save_pretrained
    from radzero_amd.weights import make_state_dict
    for i, over in enumerate(({}, {"compute_logits_type": "global_alignment", "use_text_projection": True, "sim_op": "dot"})):
        cfg = RadZeroConfig(vit_layers=1, align_layers=1, text_layers=1, vocab_size=1000, loss_temperature=0.05, **over)
        sd = make_state_dict(cfg, 5)

        class _M:
            config, _state_dict, dtype = cfg, sd, torch.float32
        d = save_pretrained(_M(), str(tmp_path / f"saved{i}"))
        assert sorted(os.listdir(d)) == ["config.json", "configuration_radzero_hip.py", "model.safetensors", "modeling_radzero_hip.py"]
        back = load_checkpoint(d)
        assert set(back) == set(sd) and all(np.array_equal(back[k], np.asarray(sd[k], np.float32)) for k in sd)
        assert config_from_hf(d, state_dict=back) == cfg
        c = AutoConfig.from_pretrained(d)
        assert isinstance(c, RadZeroHFConfig) and c.to_radzero(back) == cfg and c.dtype in ("float32", torch.float32)


def test_from_pretrained_names_what_it_cannot_load_and_warns_on_unknown_kwargs(tmp_path):
    """a hub id gets a clear error (no network resolution here), unknown keyword arguments are not swallowed silently."""
    import pytest
    from radzero_amd.hf import RadZeroHFModel
    with pytest.raises(FileNotFoundError, match="snapshot_download"):
        RadZeroHFModel.from_pretrained("Deepnoid/RadZero", torch_dtype="float32", device_map="cuda")
    cfg, src = _layout(tmp_path)
    with pytest.warns(RuntimeWarning, match="frobnicate"):
        with pytest.raises(Exception):                          # no HIP device here: construction raises AFTER the warning
            RadZeroHFModel.from_pretrained(str(src), torch_dtype="float32", device_map="cuda", frobnicate=1)
