"""The `AutoModel` surface of RadZero's README (README.md:72-89):

    model = AutoModel.from_pretrained("Deepnoid/RadZero", trust_remote_code=True, torch_dtype=dtype, device_map=device)

resolving to the HIP model.  transformers finds a custom model in two ways, and both are provided:

  * **auto_map + remote code** (what the hub repository does: config.json names `module.Class` files that live beside the weights).
    `export_auto_map(checkpoint_dir)` writes two three-line module files into a LOCAL checkpoint directory (the layout the reference's
    `save_pretrained` produces, exp/cxr_pt/model/configuration.py:107-129) and adds `auto_map` / `model_type` to its config.json; after
    that the README's call, with the directory in place of the hub id, returns a `RadZeroModel` — nothing needs to be imported first.
  * **in-process registration**: `import radzero_amd.hf` registers model_type "radzero_hip" with AutoConfig / AutoModel, so a directory whose
    config.json carries that model_type loads without `trust_remote_code`.

The reference's own config.json has no model_type (CxrAlignConfig defines none) and no auto_map: without `export_auto_map` only
`RadZeroModel.from_pretrained(dir)` can read it — AutoModel has nothing to dispatch on.  The tokenizer and the image processor of the
README (`AutoTokenizer`, `AutoImageProcessor`) are transformers' own and are not touched.
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Optional

from transformers import AutoConfig, AutoModel, PretrainedConfig

from .modeling import RadZeroModel

MODEL_TYPE = "radzero_hip"
CONFIG_MODULE, MODEL_MODULE = "configuration_radzero_hip", "modeling_radzero_hip"
AUTO_MAP = {"AutoConfig": f"{CONFIG_MODULE}.RadZeroHFConfig", "AutoModel": f"{MODEL_MODULE}.RadZeroHFModel"}

_SHIM = '"""{what} of the MI355X-native RadZero path: resolved by transformers through config.json\'s auto_map (trust_remote_code=True).\nThe implementation is the installed `radzero_amd` package (libradzero_hip.so); this file only names it."""\nfrom radzero_amd.hf import {name}  # noqa: F401\n'


class RadZeroHFConfig(PretrainedConfig):
    """config.json of a CxrAlignModel checkpoint (exp/cxr_pt/model/configuration.py:107-129) as transformers sees it: the three
    sub-configurations stay plain dicts, every other key (loss, compute_logits_type, ...) is kept as written.  `to_radzero()` is the
    kernels' view of it (radzero_amd.checkpoint.config_from_hf)."""
    model_type = MODEL_TYPE

    def __init__(self, vision_config: Optional[dict] = None, text_config: Optional[dict] = None,
                 align_transformer_config: Optional[dict] = None, **kwargs):
        self.vision_config = dict(vision_config or {})
        self.text_config = dict(text_config or {})
        self.align_transformer_config = dict(align_transformer_config or {})
        super().__init__(**kwargs)

    def to_radzero(self, state_dict=None):
        from .checkpoint import config_from_hf
        return config_from_hf(self.to_dict(), state_dict=state_dict)


class RadZeroHFModel(RadZeroModel):
    """RadZeroModel under the names transformers' auto classes look for.  Not a torch.nn.Module: `from_pretrained` builds the HIP handle."""
    config_class = RadZeroHFConfig

    @classmethod
    def register_for_auto_class(cls, auto_class="AutoModel"):      # AutoModel.from_pretrained calls it on a remote-code class
        return None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, **kwargs):
        """What `AutoModel.from_pretrained(dir, trust_remote_code=True, torch_dtype=..., device_map=...)` ends in.  `config` arrives as
        the RadZeroHFConfig transformers already parsed; the tensor shapes of the checkpoint still win over it (config_from_hf)."""
        import torch
        from .checkpoint import config_from_hf, load_checkpoint
        from .modeling import _resolve_device
        if model_args:
            raise TypeError("RadZeroHFModel.from_pretrained takes no positional model arguments")
        dtype, alt = kwargs.pop("torch_dtype", None), kwargs.pop("dtype", None)      # transformers >= 4.56 spells it `dtype`
        dtype = alt if dtype is None else dtype
        if isinstance(dtype, str):
            dtype = None if dtype == "auto" else getattr(torch, dtype.replace("torch.", ""))
        if dtype is None:               # "auto" / absent: what the checkpoint was saved in (the README always passes one)
            saved = (getattr(config, "dtype", None) or getattr(config, "torch_dtype", None)) if config is not None else None
            dtype = getattr(torch, str(saved).replace("torch.", "")) if saved else torch.float32
        device, device_map = kwargs.pop("device", None), kwargs.pop("device_map", None)
        if device is not None and device_map is not None and _resolve_device(device) != _resolve_device(device_map):
            raise ValueError("from_pretrained: `device` and `device_map` name different devices")
        dev = _resolve_device(device if device is not None else device_map)
        path = str(pretrained_model_name_or_path)
        if not os.path.exists(path):
            raise FileNotFoundError(f"RadZeroHFModel.from_pretrained: {path!r} is not a local checkpoint directory or file.  Hub ids (e.g. \"Deepnoid/RadZero\") are not "
                                    "resolved here: download the repository first (huggingface_hub.snapshot_download) and pass its directory.")
        known = {"trust_remote_code", "low_cpu_mem_usage", "attn_implementation", "revision", "cache_dir", "local_files_only", "token", "use_safetensors",
                 "force_download", "proxies", "subfolder", "_from_auto", "_commit_hash", "adapter_kwargs", "code_revision", "name_or_path", "_name_or_path", "use_auth_token", "weights_only"}
        extra = sorted(k for k in kwargs if k not in known)
        if extra:
            import warnings
            warnings.warn(f"RadZeroHFModel.from_pretrained: ignoring unexpected keyword argument(s) {extra}", RuntimeWarning, stacklevel=2)
        sd = load_checkpoint(path)
        if isinstance(config, PretrainedConfig):
            rz_cfg = config_from_hf(config.to_dict(), state_dict=sd)
        elif config is not None:
            rz_cfg = config                                  # a RadZeroConfig
        else:
            has_cfg = os.path.isdir(path) and os.path.exists(os.path.join(path, "config.json"))
            rz_cfg = config_from_hf(path if has_cfg else {}, state_dict=sd)
        return cls.from_state_dict(sd, rz_cfg, torch_dtype=dtype, device=dev).eval()


def config_dict(cfg, dtype=None) -> dict:
    """A RadZeroConfig in the LAYOUT of the reference's config.json (CxrAlignConfig, exp/cxr_pt/model/configuration.py:107-129: three nested
    configurations + the loss / head keys at the top level), with the auto_map that makes it loadable through AutoModel.  Inverse of
    checkpoint.config_from_hf for the fields the kernels read."""
    return {
        "architectures": ["RadZeroHFModel"], "model_type": MODEL_TYPE, "auto_map": dict(AUTO_MAP),
        "vision_config": {"model_type": "dinov2", "hidden_size": cfg.hidden_size, "num_attention_heads": cfg.num_attention_heads, "mlp_ratio": cfg.mlp_ratio,
                          "patch_size": cfg.patch_size, "num_channels": cfg.num_channels, "image_size": cfg.pretrain_image_size,
                          "num_hidden_layers": cfg.vit_layers, "layer_norm_eps": cfg.vit_layer_norm_eps},
        "text_config": {"model_type": "mpnet", "vocab_size": cfg.vocab_size, "max_position_embeddings": cfg.max_position_embeddings,
                        "num_hidden_layers": cfg.text_layers, "intermediate_size": cfg.text_intermediate_size, "layer_norm_eps": cfg.text_layer_norm_eps,
                        "relative_attention_num_buckets": cfg.relative_attention_num_buckets, "use_text_projection": bool(cfg.use_text_projection)},
        "align_transformer_config": {"model_type": "align_transformer", "num_hidden_layers": cfg.align_layers, "use_layer_norm": False},
        "loss": {"RadZeroLoss": {"loss_temperature": cfg.loss_temperature, "sim_op": cfg.sim_op, "attn_temperature": cfg.attn_temperature,
                                 "hidden_dim": cfg.hidden_size}, "apply": ["RadZeroLoss"], "ratio": [1.0]},
        "compute_logits_type": cfg.compute_logits_type,
        **({"dtype": str(dtype).replace("torch.", "")} if dtype is not None else {}),
    }


def save_pretrained(model, save_directory: str) -> str:
    """`model.save_pretrained(dir)` for the HIP model: model.safetensors with the reference's tensor names (the state dict the model was
    loaded from), config.json in the reference's layout + auto_map, and the two module files — a directory that both
    `RadZeroModel.from_pretrained` and the README's `AutoModel.from_pretrained(dir, trust_remote_code=True, ...)` read back."""
    from .checkpoint import save_checkpoint
    sd = getattr(model, "_state_dict", None)
    if sd is None:
        raise RuntimeError("save_pretrained: the model holds no state dict (load weights first)")
    os.makedirs(save_directory, exist_ok=True)
    import torch
    save_checkpoint({k: (v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}, save_directory)
    with open(os.path.join(save_directory, "config.json"), "w") as f:
        json.dump(config_dict(model.config, getattr(model, "dtype", None)), f, indent=2, sort_keys=True)
    with open(os.path.join(save_directory, CONFIG_MODULE + ".py"), "w") as f:
        f.write(_SHIM.format(what="Configuration class", name="RadZeroHFConfig"))
    with open(os.path.join(save_directory, MODEL_MODULE + ".py"), "w") as f:
        f.write(_SHIM.format(what="Model class", name="RadZeroHFModel"))
    return save_directory


RadZeroModel.save_pretrained = lambda self, save_directory, **_ignored: save_pretrained(self, save_directory)


def register() -> None:
    """model_type "radzero_hip" -> (RadZeroHFConfig, RadZeroHFModel) in transformers' auto classes (idempotent)."""
    AutoConfig.register(MODEL_TYPE, RadZeroHFConfig, exist_ok=True)
    AutoModel.register(RadZeroHFConfig, RadZeroHFModel, exist_ok=True)


def export_auto_map(checkpoint_dir: str, out_dir: Optional[str] = None, link_weights: bool = True, in_place: bool = False) -> str:
    """Make a CxrAlignModel checkpoint directory loadable by the README's `AutoModel.from_pretrained(dir, trust_remote_code=True, ...)`
    as the HIP model.  Writes into `out_dir` (weights symlinked, or copied with link_weights=False): the original stays untouched and the
    reference's own loader keeps reading it.  `in_place=True` (explicit: it rewrites the checkpoint's config.json — model_type, architectures,
    auto_map — after which `CxrAlignConfig.from_pretrained` sees a foreign model_type) modifies `checkpoint_dir` itself.  Returns the directory to
    pass to from_pretrained."""
    src = os.path.abspath(checkpoint_dir)
    cfg_path = os.path.join(src, "config.json")
    if not os.path.isfile(cfg_path):
        raise FileNotFoundError(f"{cfg_path}: not a save_pretrained directory")
    if out_dir is None and not in_place:
        raise ValueError("export_auto_map: pass out_dir= (the original checkpoint stays untouched) or in_place=True to rewrite the checkpoint's own config.json")
    if out_dir is not None and in_place and os.path.abspath(out_dir) != src:
        raise ValueError("export_auto_map: out_dir and in_place=True name two different targets")
    dst = os.path.abspath(out_dir) if out_dir else src
    if dst != src:
        os.makedirs(dst, exist_ok=True)
        for name in os.listdir(src):
            if name == "config.json" or name.endswith(".py"):
                continue
            s, d = os.path.join(src, name), os.path.join(dst, name)
            if os.path.lexists(d):
                os.remove(d)
            if link_weights and os.path.isfile(s):
                os.symlink(s, d)
            elif os.path.isfile(s):
                shutil.copy2(s, d)
    cfg = json.load(open(cfg_path))
    cfg["auto_map"] = dict(AUTO_MAP)
    cfg["model_type"] = MODEL_TYPE
    cfg["architectures"] = ["RadZeroHFModel"]
    tmp = os.path.join(dst, f"config.json.tmp{os.getpid()}")
    with open(tmp, "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    os.replace(tmp, os.path.join(dst, "config.json"))
    with open(os.path.join(dst, CONFIG_MODULE + ".py"), "w") as f:
        f.write(_SHIM.format(what="Configuration class", name="RadZeroHFConfig"))
    with open(os.path.join(dst, MODEL_MODULE + ".py"), "w") as f:
        f.write(_SHIM.format(what="Model class", name="RadZeroHFModel"))
    return dst


register()
