"""Import the reference CxrAlignModel in THIS container (never on the GPU box).

Recipe from SURVEY.md §8(c): the package `exp.cxr_pt.model` cannot be imported as-is (peft,
open_clip, common.trainer … are absent — ordinary ImportErrors), so synthetic package modules
point at the reference directories, the three unused third-party imports are stubbed, and the
three network calls (`from_pretrained`) are replaced by config-built random-init modules.
Nothing from /root/reference is copied; the modules are imported from where they lie.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF = os.environ.get("RADZERO_REFERENCE", "/root/reference")


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def load_reference_model(cfg, state_dict=None, attn_implementation="eager"):      # cfg.compute_logits_type / cfg.use_text_projection select the alignment heads
    """Build the reference `CxrAlignModel` (fp32, eval) for `cfg` (radzero_amd.config.RadZeroConfig)."""
    import logging

    import torch
    from transformers import Dinov2Config, Dinov2Model, MPNetConfig, MPNetModel

    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "exp.cxr_pt.model.modeling" not in sys.modules:
        _pkg("exp", os.path.join(REF, "exp"))
        _pkg("exp.cxr_pt", os.path.join(REF, "exp/cxr_pt"))
        _pkg("exp.cxr_pt.model", os.path.join(REF, "exp/cxr_pt/model"))
        # unused-on-this-path third-party / sibling imports (losses.py:7, vision_encoders.py:12-14)
        oc = _pkg("open_clip")
        ocl = _pkg("open_clip.loss")
        ocl.ClipLoss = type("ClipLoss", (torch.nn.Module,), {})
        ocl.SigLipLoss = type("SigLipLoss", (torch.nn.Module,), {})
        oc.loss = ocl
        _pkg("common", os.path.join(REF, "common"))
        ct = _pkg("common.trainer")
        ct.logger = logging.getLogger("reference")
        _pkg("external")
        _pkg("external.CARZero")
        _pkg("external.CARZero.CARZero")
        _pkg("external.CARZero.CARZero.models")
        tb = _pkg("external.CARZero.CARZero.models.transformer_backbones")
        tb.MRM = type("MRM", (torch.nn.Module,), {})
        tb.load_weight = lambda *a, **k: None
        for mod in ("configuration", "losses", "align_transformers", "text_encoders",
                    "vision_encoders", "common_layers", "modeling"):
            importlib.import_module(f"exp.cxr_pt.model.{mod}")

    configuration = sys.modules["exp.cxr_pt.model.configuration"]
    vision_encoders = sys.modules["exp.cxr_pt.model.vision_encoders"]
    text_encoders = sys.modules["exp.cxr_pt.model.text_encoders"]
    modeling = sys.modules["exp.cxr_pt.model.modeling"]

    dcfg = Dinov2Config(
        hidden_size=cfg.hidden_size, num_hidden_layers=cfg.vit_layers,
        num_attention_heads=cfg.num_attention_heads, mlp_ratio=cfg.mlp_ratio,
        image_size=cfg.pretrain_image_size, patch_size=cfg.patch_size,
        layer_norm_eps=cfg.vit_layer_norm_eps)
    dcfg._attn_implementation = attn_implementation
    mcfg = MPNetConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.text_layers,
        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.text_intermediate_size,
        max_position_embeddings=cfg.max_position_embeddings, layer_norm_eps=cfg.text_layer_norm_eps,
        relative_attention_num_buckets=cfg.relative_attention_num_buckets)
    mcfg._attn_implementation = "eager"          # MPNet has no SDPA path in transformers; `attn_implementation` selects the vision side only

    class _AutoConfig:                      # configuration.py:25-27
        @staticmethod
        def from_pretrained(name, *a, **k):
            return dcfg

    class _Dinov2:                          # vision_encoders.py:29
        @staticmethod
        def from_pretrained(name, *a, **k):
            return Dinov2Model(dcfg)

    class _AutoModel:                       # text_encoders.py:14
        @staticmethod
        def from_pretrained(name, *a, **k):
            return MPNetModel(mcfg)

    configuration.AutoConfig = _AutoConfig
    vision_encoders.Dinov2Model = _Dinov2
    text_encoders.AutoModel = _AutoModel
    modeling.Dinov2Model = Dinov2Model      # keep the real class for isinstance (modeling.py:98)

    # exp/cxr_pt/configs/radzero.yaml:14-48, verbatim hyper-parameters
    model_config = dict(
        vision_config=dict(model_type="dinov2",
                           pretrained_name_or_path="StanfordAIMI/dinov2-base-xray-224", img_size=518),
        text_config=dict(use_text_projection=bool(getattr(cfg, "use_text_projection", False)), model_type="mpnet",
                         pretrained_name_or_path="sentence-transformers/all-mpnet-base-v2",
                         pretrained_tokenizer_name_or_path="sentence-transformers/all-mpnet-base-v2",
                         use_cls_token=False),
        align_transformer_config=dict(model_type="align_transformer", hidden_size=cfg.hidden_size,
                                      num_hidden_layers=cfg.align_layers, projector_config=None,
                                      use_layer_norm=False),
        loss=dict(apply=["RadZeroLoss"], ratio=[1.0],
                  RadZeroLoss=dict(hidden_dim=cfg.hidden_size, mpnce_row_sum=False, mpnce_col_sum=False,
                                   attn_temperature=getattr(cfg, "attn_temperature", None), loss_temperature=cfg.loss_temperature,
                                   text_features_l2_norm=False, sim_op=cfg.sim_op)),
        compute_logits_type=getattr(cfg, "compute_logits_type", "radzero"),
        pretrained_dir="/data/pretrained",          # exp/cxr_pt/configs/paths.yaml:11 (only read for m3ae)
    )
    rcfg = configuration.CxrAlignConfig(**model_config)
    rcfg.align_transformer_config._attn_implementation = "eager"     # the reference's AlignTransformer declares no SDPA support
    model = modeling.CxrAlignModel(rcfg).eval().float()
    # the released compute_logits reads an attribute __init__ never sets
    # (modeling.py:320); the only self-consistent branch is compute_i2t_loss == False.
    model.loss_fns["RadZeroLoss"].compute_i2t_loss = False
    if state_dict is not None:
        sd = {k: torch.from_numpy(v) for k, v in state_dict.items()}
        ref_keys = set(model.state_dict().keys())
        missing = ref_keys - set(sd)
        extra = set(sd) - ref_keys
        assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
        model.load_state_dict(sd, strict=True)
    return model
