"""Checkpoint import: read a RadZero / CxrAlignModel checkpoint directory (or file) written by
HF `save_pretrained` (`model.safetensors` / `pytorch_model.bin`, optionally sharded with an index json) into the
name -> fp32 array dict `RadZeroModel.load_state_dict` takes.  The tensor names are the reference's own
(exp/cxr_pt/model/modeling.py:55-86); a leading "model." / "module." prefix (Trainer / DDP wrappers) is stripped.
`config_from_hf` reads the hyper-parameters the kernels need from the checkpoint's config.json
(exp/cxr_pt/model/configuration.py:107-129) instead of assuming the released values.
"""
from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np

from .config import RadZeroConfig

_PREFIXES = ("model.", "module.", "base_model.model.")


def _strip(name: str) -> str:
    changed = True
    while changed:
        changed = False
        for p in _PREFIXES:
            if name.startswith(p):
                name, changed = name[len(p):], True
    return name


def _read_file(path: str) -> Dict[str, np.ndarray]:
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        out = {}
        with safe_open(path, framework="pt") as f:           # "pt": bf16 checkpoints have no numpy dtype
            for k in f.keys():
                out[k] = f.get_tensor(k).float().numpy()
        return out
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k: v.float().numpy() for k, v in sd.items() if hasattr(v, "float")}


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """`path`: a checkpoint directory, or a .safetensors / .bin / .pt file."""
    files = []
    if os.path.isdir(path):
        for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
            ip = os.path.join(path, index)
            if os.path.exists(ip):
                shards = sorted(set(json.load(open(ip))["weight_map"].values()))
                files = [os.path.join(path, s) for s in shards]
                break
        if not files:
            for single in ("model.safetensors", "pytorch_model.bin"):
                if os.path.exists(os.path.join(path, single)):
                    files = [os.path.join(path, single)]
                    break
        if not files:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin (or index) under {path}")
    else:
        files = [path]
    out: Dict[str, np.ndarray] = {}
    for f in files:
        for k, v in _read_file(f).items():
            out[_strip(k)] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def save_checkpoint(state_dict, path: str) -> str:
    """Write `model.safetensors` under `path` with the reference's tensor names (round-trip / test helper)."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    f = os.path.join(path, "model.safetensors")
    save_file({k: np.ascontiguousarray(np.asarray(v, np.float32)) for k, v in state_dict.items()}, f)
    return f


def _shape_facts(state_dict) -> dict:
    """Hyper-parameters that a CxrAlignModel checkpoint only carries as tensor shapes / key counts.  The reference's
    config.json (`CxrAlignConfig.save_pretrained`, configuration.py:107-129 — fixture tests/golden/hf_layout/config.json)
    stores the text side as `pretrained_name_or_path` only: vocabulary, depth, widths of the MPNet encoder are whatever
    `AutoModel.from_pretrained(name)` built (text_encoders.py:13-14), i.e. visible in the state dict and nowhere else."""
    facts = {}
    if not state_dict:
        return facts
    def depth(prefix):
        idx = {int(k[len(prefix):].split(".", 1)[0]) for k in state_dict if k.startswith(prefix)}
        return (max(idx) + 1) if idx else None
    shape = lambda k: tuple(np.shape(state_dict[k])) if k in state_dict else None
    # a layer count is a fact only when the checkpoint has at least one key of that stack: a vision-only or partial state dict
    # says nothing about the others (config.json / the defaults keep speaking for them)
    for field, prefix in (("vit_layers", "vision_model.encoder.layer."), ("align_layers", "align_transformer.transformer_layers.layer."),
                          ("text_layers", "text_model.encoder.layer.")):
        n = depth(prefix)
        if n is not None:
            facts[field] = n
    we = shape("text_model.embeddings.word_embeddings.weight")
    pe = shape("text_model.embeddings.position_embeddings.weight")
    rb = shape("text_model.encoder.relative_attention_bias.weight")
    inter = shape("text_model.encoder.layer.0.intermediate.dense.weight")
    vpos = shape("vision_model.embeddings.position_embeddings")
    cls = shape("vision_model.embeddings.cls_token")
    pw = shape("vision_model.embeddings.patch_embeddings.projection.weight")
    fc1 = shape("vision_model.encoder.layer.0.mlp.fc1.weight")
    # the shared width and the head count are VISION-side fields of RadZeroConfig: they come from vision tensors.  The text side has
    # to agree (one `hidden_size` / `num_attention_heads` serves both encoders and the VL-CABS head): a mismatch is an error, not
    # something to overwrite silently.
    vis_hidden = (pw[0] if pw else None) or (cls[-1] if cls else None)
    if vis_hidden: facts["hidden_size"] = vis_hidden
    if fc1 and vis_hidden and fc1[1] == vis_hidden and fc1[0] % vis_hidden == 0: facts["mlp_ratio"] = fc1[0] // vis_hidden
    if we:
        facts["vocab_size"] = we[0]
        if vis_hidden and we[1] != vis_hidden:
            raise ValueError(f"text embedding width {we[1]} differs from the vision width {vis_hidden}: not a CxrAlignModel checkpoint this path supports")
        if not vis_hidden: facts["hidden_size"] = we[1]
    if pe: facts["max_position_embeddings"] = pe[0]
    if rb: facts["relative_attention_num_buckets"], facts["text_num_attention_heads"] = rb
    if inter: facts["text_intermediate_size"] = inter[0]
    if pw: facts["num_channels"], facts["patch_size"] = pw[1], pw[2]
    if vpos and pw:
        facts["pretrain_image_size"] = int(round((vpos[-2] - 1) ** 0.5)) * pw[2]
    return facts


def config_from_hf(path_or_dict, state_dict=None) -> RadZeroConfig:
    """RadZeroConfig from a CxrAlignConfig config.json (vision_config / text_config / align_transformer_config and the
    loss kwargs, which `PretrainedConfig` writes at the top level — a "kwargs" nesting is accepted too) plus, when
    `state_dict` is given, the shapes of the checkpoint itself (see _shape_facts; they win where both speak, because the
    weights are what will be multiplied).  Absent fields keep the released defaults — in particular MPNet's
    layer_norm_eps, which lives in neither place (it is the hub config of all-mpnet-base-v2)."""
    d = path_or_dict
    if isinstance(d, str):
        p = os.path.join(d, "config.json") if os.path.isdir(d) else d
        d = json.load(open(p))
    v = d.get("vision_config", {}) or {}
    t = d.get("text_config", {}) or {}
    a = d.get("align_transformer_config", {}) or {}
    extra = d.get("kwargs") if isinstance(d.get("kwargs"), dict) else d
    loss = ((extra.get("loss", {}) or {}).get("RadZeroLoss", {}) or {})
    base = RadZeroConfig()
    fields = dict(
        hidden_size=v.get("hidden_size", base.hidden_size),
        num_attention_heads=v.get("num_attention_heads", base.num_attention_heads),
        mlp_ratio=v.get("mlp_ratio", base.mlp_ratio),
        patch_size=v.get("patch_size", base.patch_size),
        num_channels=v.get("num_channels", base.num_channels),
        pretrain_image_size=v.get("image_size", base.pretrain_image_size),
        vit_layers=v.get("num_hidden_layers", base.vit_layers),
        vit_layer_norm_eps=v.get("layer_norm_eps", base.vit_layer_norm_eps),
        align_layers=a.get("num_hidden_layers", base.align_layers),
        vocab_size=t.get("vocab_size", base.vocab_size),
        max_position_embeddings=t.get("max_position_embeddings", base.max_position_embeddings),
        text_layers=t.get("num_hidden_layers", base.text_layers),
        text_intermediate_size=t.get("intermediate_size", base.text_intermediate_size),
        text_layer_norm_eps=t.get("layer_norm_eps", base.text_layer_norm_eps),
        relative_attention_num_buckets=t.get("relative_attention_num_buckets", base.relative_attention_num_buckets),
        loss_temperature=loss.get("loss_temperature", base.loss_temperature),
        sim_op=loss.get("sim_op", base.sim_op),
        attn_temperature=loss.get("attn_temperature", base.attn_temperature),
    )
    facts = _shape_facts(state_dict)
    text_heads = facts.pop("text_num_attention_heads", None)          # MPNet's head count (relative_attention_bias columns)
    fields.update(facts)
    if text_heads is not None and text_heads != fields["num_attention_heads"]:
        raise ValueError(f"MPNet has {text_heads} attention heads, the vision encoder {fields['num_attention_heads']}: the kernels share one head count")
    cfg = RadZeroConfig(**fields)
    if a.get("use_layer_norm"):
        raise NotImplementedError("align_transformer_config.use_layer_norm=True is not part of the released model")
    cfg.use_text_projection = bool(t.get("use_text_projection"))
    cfg.compute_logits_type = extra.get("compute_logits_type") or "radzero"
    if cfg.compute_logits_type not in ("radzero", "cls_alignment", "global_alignment"):
        raise NotImplementedError(f"compute_logits_type {cfg.compute_logits_type!r} (modeling.py:288-353 knows radzero / cls_alignment / global_alignment)")
    return cfg
