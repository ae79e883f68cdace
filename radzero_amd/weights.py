"""Seeded synthetic checkpoint in the reference's state_dict naming.

Pretrained `Deepnoid/RadZero` weights are unreachable offline, so parity and
benchmarks run on a deterministic synthetic checkpoint.  The key names and shapes are the ones
`CxrAlignModel` registers (exp/cxr_pt/model/modeling.py:55-86; align_transformers.py:28;
losses.py:51-56) so that the same dict loads into the reference model (tools/make_goldens.py does
exactly that) and into `radzero_amd.modeling.RadZeroModel.from_state_dict`.

Values come from numpy's Philox counter RNG keyed by (seed, crc32(name)): bit-identical on any
machine / numpy version, independent of generation order.  Statistics are chosen "trained-like"
rather than init-like (non-zero biases, LN gains != 1, LayerScale spread over [0.05, 0.6],
peaky-ish attention) so that every term of every kernel is exercised.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np

from .config import RadZeroConfig


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std, mean=0.0):
    x = _rng(seed, name).standard_normal(size=shape, dtype=np.float32)
    return (x * np.float32(std) + np.float32(mean)).astype(np.float32)


def _uniform(seed, name, shape, lo, hi):
    x = _rng(seed, name).random(size=shape, dtype=np.float32)
    return (x * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def checkpoint_spec(cfg: RadZeroConfig) -> "OrderedDict[str, tuple]":
    """name -> (shape, kind).  `kind` selects the value distribution in `make_state_dict`."""
    d, f = cfg.hidden_size, cfg.intermediate_size
    p = cfg.patch_size
    g0 = cfg.pretrain_image_size // p
    spec: "OrderedDict[str, tuple]" = OrderedDict()

    def dino_block(prefix):
        spec[f"{prefix}.norm1.weight"] = ((d,), "ln_w")
        spec[f"{prefix}.norm1.bias"] = ((d,), "ln_b")
        for n in ("query", "key", "value"):
            spec[f"{prefix}.attention.attention.{n}.weight"] = ((d, d), "qk_w" if n != "value" else "lin_w")
            spec[f"{prefix}.attention.attention.{n}.bias"] = ((d,), "bias")
        spec[f"{prefix}.attention.output.dense.weight"] = ((d, d), "lin_w")
        spec[f"{prefix}.attention.output.dense.bias"] = ((d,), "bias")
        spec[f"{prefix}.layer_scale1.lambda1"] = ((d,), "layerscale")
        spec[f"{prefix}.norm2.weight"] = ((d,), "ln_w")
        spec[f"{prefix}.norm2.bias"] = ((d,), "ln_b")
        spec[f"{prefix}.mlp.fc1.weight"] = ((f, d), "lin_w")
        spec[f"{prefix}.mlp.fc1.bias"] = ((f,), "bias")
        spec[f"{prefix}.mlp.fc2.weight"] = ((d, f), "lin_w_wide")
        spec[f"{prefix}.mlp.fc2.bias"] = ((d,), "bias")
        spec[f"{prefix}.layer_scale2.lambda1"] = ((d,), "layerscale")

    # --- vision_model: transformers Dinov2Model ---
    spec["vision_model.embeddings.cls_token"] = ((1, 1, d), "emb")
    spec["vision_model.embeddings.mask_token"] = ((1, d), "zeros")
    spec["vision_model.embeddings.position_embeddings"] = ((1, g0 * g0 + 1, d), "emb")
    spec["vision_model.embeddings.patch_embeddings.projection.weight"] = ((d, cfg.num_channels, p, p), "patch_w")
    spec["vision_model.embeddings.patch_embeddings.projection.bias"] = ((d,), "bias")
    for i in range(cfg.vit_layers):
        dino_block(f"vision_model.encoder.layer.{i}")
    spec["vision_model.layernorm.weight"] = ((d,), "ln_w")
    spec["vision_model.layernorm.bias"] = ((d,), "ln_b")
    # --- align_transformer: Dinov2Encoder (align_transformers.py:28) ---
    for i in range(cfg.align_layers):
        dino_block(f"align_transformer.transformer_layers.layer.{i}")
    # --- text_model: transformers MPNetModel ---
    spec["text_model.embeddings.word_embeddings.weight"] = ((cfg.vocab_size, d), "emb_tok")
    spec["text_model.embeddings.position_embeddings.weight"] = ((cfg.max_position_embeddings, d), "emb_tok")
    spec["text_model.embeddings.LayerNorm.weight"] = ((d,), "ln_w")
    spec["text_model.embeddings.LayerNorm.bias"] = ((d,), "ln_b")
    ft = cfg.text_intermediate_size
    for i in range(cfg.text_layers):
        pre = f"text_model.encoder.layer.{i}"
        for n in ("q", "k", "v", "o"):
            spec[f"{pre}.attention.attn.{n}.weight"] = ((d, d), "qk_w" if n in "qk" else "lin_w")
            spec[f"{pre}.attention.attn.{n}.bias"] = ((d,), "bias")
        spec[f"{pre}.attention.LayerNorm.weight"] = ((d,), "ln_w")
        spec[f"{pre}.attention.LayerNorm.bias"] = ((d,), "ln_b")
        spec[f"{pre}.intermediate.dense.weight"] = ((ft, d), "lin_w")
        spec[f"{pre}.intermediate.dense.bias"] = ((ft,), "bias")
        spec[f"{pre}.output.dense.weight"] = ((d, ft), "lin_w_wide")
        spec[f"{pre}.output.dense.bias"] = ((d,), "bias")
        spec[f"{pre}.output.LayerNorm.weight"] = ((d,), "ln_w")
        spec[f"{pre}.output.LayerNorm.bias"] = ((d,), "ln_b")
    spec["text_model.encoder.relative_attention_bias.weight"] = (
        (cfg.relative_attention_num_buckets, cfg.num_attention_heads), "relbias")
    spec["text_model.pooler.dense.weight"] = ((d, d), "lin_w")     # computed-but-unused by the path
    spec["text_model.pooler.dense.bias"] = ((d,), "bias")
    if getattr(cfg, "use_text_projection", False):                 # text_projector = nn.Linear(text_dim, 2 * hidden) (modeling.py:70-73)
        spec["text_projector.weight"] = ((2 * d, d), "lin_w")
        spec["text_projector.bias"] = ((2 * d,), "bias")
    # --- loss_fns.RadZeroLoss (losses.py:51-56) ---
    spec["loss_fns.RadZeroLoss.layer_norm.weight"] = ((d,), "ln_w")
    spec["loss_fns.RadZeroLoss.layer_norm.bias"] = ((d,), "ln_b")
    spec["loss_fns.RadZeroLoss.loss_temperature"] = ((1,), "log_temperature")
    if getattr(cfg, "attn_temperature", None) is not None:          # the Parameter exists only when the config sets it (losses.py:57-63)
        spec["loss_fns.RadZeroLoss.attn_temperature"] = ((1,), "log_attn_temperature")
    return spec


def _make(seed: int, name: str, shape, kind: str, cfg: RadZeroConfig) -> np.ndarray:
    d = cfg.hidden_size
    if kind == "zeros":
        return np.zeros(shape, np.float32)
    if kind == "ln_w":
        return _normal(seed, name, shape, 0.15, 1.0)
    if kind == "ln_b":
        return _normal(seed, name, shape, 0.05)
    if kind == "bias":
        return _normal(seed, name, shape, 0.05)
    if kind == "lin_w":
        return _normal(seed, name, shape, 1.0 / math.sqrt(d))
    if kind == "qk_w":
        return _normal(seed, name, shape, 1.6 / math.sqrt(d))
    if kind == "lin_w_wide":
        return _normal(seed, name, shape, 1.0 / math.sqrt(shape[1]))
    if kind == "patch_w":
        return _normal(seed, name, shape, 1.0 / math.sqrt(shape[1] * shape[2] * shape[3]))
    if kind == "emb":
        return _normal(seed, name, shape, 0.3)
    if kind == "emb_tok":
        return _normal(seed, name, shape, 0.5)
    if kind == "layerscale":
        return _uniform(seed, name, shape, 0.05, 0.6)
    if kind == "relbias":
        return _normal(seed, name, shape, 0.7)
    if kind == "log_temperature":
        return np.array([math.log(cfg.loss_temperature)], np.float32)
    if kind == "log_attn_temperature":
        return np.array([math.log(cfg.attn_temperature)], np.float32)
    raise KeyError(kind)


def make_state_dict(cfg: RadZeroConfig | None = None, seed: int = 20260103,
                    prefixes: tuple | None = None) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic checkpoint (numpy fp32) in the reference's naming.

    `prefixes` restricts generation to names starting with one of them (e.g. only `text_model.`)."""
    cfg = cfg or RadZeroConfig()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, kind) in checkpoint_spec(cfg).items():
        if prefixes is not None and not name.startswith(tuple(prefixes)):
            continue
        out[name] = _make(seed, name, shape, kind, cfg)
    return out


def add_outlier_channels(sd, cfg: RadZeroConfig | None = None, seed: int = 77, n_channels: int = 4,
                         ln_gain: float = 0.08, fc2_gain: float = 400.0):
    """Copy of `sd` with "massive activation" channels, the feature of trained DINOv2 checkpoints the benign generator
    above lacks: `n_channels` fixed hidden channels get their fc2 output rows (+ bias) multiplied by `fc2_gain` in every
    Dinov2 block, so the fp32 residual stream carries a few values two orders of magnitude above the rest and every
    LayerNorm's mean / variance is dominated by them; as in trained checkpoints, the LN gains of those channels are small
    (`ln_gain`), which keeps the network well conditioned (with LARGE gains on them even two fp32 implementations
    disagree by > 1 on the scores: attention logits of 1e4 make the soft-max a chaotic arg-max).  Used to test the 16-bit
    modes' range and precision behaviour (tests/golden/g8_outlier_*.npz holds the reference's outputs for it)."""
    cfg = cfg or RadZeroConfig()
    ch = np.sort(_rng(seed, "outlier_channels").choice(cfg.hidden_size, size=n_channels, replace=False))
    out = OrderedDict((k, np.array(v, np.float32, copy=True)) for k, v in sd.items())
    for k in out:
        if not (k.startswith("vision_model.") or k.startswith("align_transformer.")):
            continue
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "vision_model.layernorm.weight":
            out[k][ch] *= np.float32(ln_gain)
        elif k.endswith("mlp.fc2.weight"):
            out[k][ch, :] *= np.float32(fc2_gain)
        elif k.endswith("mlp.fc2.bias"):
            out[k][ch] *= np.float32(fc2_gain)
    return out


def state_dict_digest(sd) -> str:
    """Order-independent fingerprint of a checkpoint (used to pin goldens to the generator)."""
    acc = 0
    for name in sorted(sd):
        a = np.ascontiguousarray(np.asarray(sd[name], dtype=np.float32))
        acc ^= zlib.crc32(a.tobytes(), zlib.crc32(name.encode()))
    return f"{acc:08x}"
