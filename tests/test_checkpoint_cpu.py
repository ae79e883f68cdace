"""CPU tests of the checkpoint importer (reference tensor names, prefix stripping, sharded index, config.json)."""
import json
import os

import numpy as np
import pytest
import torch

from radzero_amd.checkpoint import config_from_hf, load_checkpoint, save_checkpoint
from radzero_amd.config import RadZeroConfig
from radzero_amd.weights import checkpoint_spec, make_state_dict


@pytest.fixture(scope="module")
def small_sd():
    cfg = RadZeroConfig(vit_layers=1, align_layers=1, text_layers=1, vocab_size=64, max_position_embeddings=40)
    return cfg, make_state_dict(cfg, 3)


def test_safetensors_round_trip(tmp_path, small_sd):
    cfg, sd = small_sd
    save_checkpoint(sd, str(tmp_path))
    back = load_checkpoint(str(tmp_path))
    assert set(back) == set(sd) == set(checkpoint_spec(cfg))
    assert all(np.array_equal(back[k], sd[k]) for k in sd)


def test_bin_with_prefix_and_bf16(tmp_path, small_sd):
    _, sd = small_sd
    tsd = {"module.model." + k: torch.from_numpy(v).to(torch.bfloat16) for k, v in sd.items()}
    torch.save(tsd, tmp_path / "pytorch_model.bin")
    back = load_checkpoint(str(tmp_path))
    assert set(back) == set(sd)
    k = "vision_model.encoder.layer.0.mlp.fc1.weight"
    assert back[k].dtype == np.float32 and np.abs(back[k] - sd[k]).max() <= 2 ** -8 * np.abs(sd[k]).max()


def test_sharded_index(tmp_path, small_sd):
    from safetensors.numpy import save_file
    _, sd = small_sd
    names = sorted(sd)
    a, b = names[: len(names) // 2], names[len(names) // 2:]
    save_file({k: sd[k] for k in a}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in b}, str(tmp_path / "model-00002-of-00002.safetensors"))
    wm = {k: "model-00001-of-00002.safetensors" for k in a}
    wm.update({k: "model-00002-of-00002.safetensors" for k in b})
    json.dump({"weight_map": wm}, open(tmp_path / "model.safetensors.index.json", "w"))
    back = load_checkpoint(str(tmp_path))
    assert set(back) == set(sd) and all(np.array_equal(back[k], sd[k]) for k in sd)


def test_missing_checkpoint(tmp_path):
    with pytest.raises(FileNotFoundError):
        load_checkpoint(str(tmp_path))


def test_config_from_hf(tmp_path):
    d = {"vision_config": {"hidden_size": 768, "num_hidden_layers": 12, "image_size": 224, "layer_norm_eps": 1e-6, "patch_size": 14},
         "text_config": {"vocab_size": 30527, "max_position_embeddings": 514, "layer_norm_eps": 1e-5, "num_hidden_layers": 12},
         "align_transformer_config": {"num_hidden_layers": 2, "use_layer_norm": False},
         "kwargs": {"loss": {"RadZeroLoss": {"loss_temperature": 0.05, "sim_op": "cos"}}, "compute_logits_type": "radzero"}}
    json.dump(d, open(tmp_path / "config.json", "w"))
    cfg = config_from_hf(str(tmp_path))
    assert cfg.align_layers == 2 and cfg.loss_temperature == 0.05 and cfg.vit_layer_norm_eps == 1e-6
    d["kwargs"]["compute_logits_type"] = "cls_alignment"
    with pytest.raises(NotImplementedError):
        config_from_hf(d)
