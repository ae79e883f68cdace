"""Host-side mirror of the reference's model object for the VL-CABS inference path.

`RadZeroModel` presents the protocol every in-repo caller of the reference uses
(exp/cxr_pt/inference/utils.py:95-100, grounding_utils.py:58-62, visualization/attention_map_base.py:32):
`compute_logits(pixel_values, [encoded_key_phrases], **ignored)`, `forward_vision_model`,
`forward_text_model`, `.device`, `.eval()`, `.to()`; the arithmetic runs in libradzero_hip.so
(include/radzero_hip.h) — PyTorch is used here only for device memory, streams and tiny host-side tables.
Reference: exp/cxr_pt/model/modeling.py:22-356 (CxrAlignModel), losses.py:33-240.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .config import RadZeroConfig

_DTYPES = {torch.float32: _lib.RZ_F32, torch.bfloat16: _lib.RZ_BF16, torch.float16: _lib.RZ_F16}


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def interpolate_pos_encoding(position_embeddings: torch.Tensor, grid_h: int, grid_w: int, mode: str = "size") -> torch.Tensor:
    """One-time host op per resolution: bicubic resize of the stored patch position grid, CLS kept
    (TF:dinov2/modeling_dinov2.py:57-95).  Returns (1 + grid_h*grid_w, D) fp32 on the CPU.

    mode "size": `F.interpolate(size=(gh, gw))`, what transformers >= 4.45 (incl. the 5.x installed here, which generated the
    goldens) does.  mode "scale_factor_0p1": the form of the reference's PINNED transformers 4.39.3 (requirements.txt:247) —
    `scale_factor=((gh + 0.1) / g0, (gw + 0.1) / g0)`, i.e. the same output size but source coordinates scaled by
    g0 / (g + 0.1) instead of g0 / g (restated from the published 4.39 source; that version is not installed here, so this
    branch is pinned only by its own properties in tests/test_checkpoint_cpu.py).  The kernels take the finished table either way."""
    pe = position_embeddings.detach().float().cpu()
    if pe.dim() == 3:
        pe = pe[0]
    npos = pe.shape[0] - 1
    g0 = int(round(math.sqrt(npos)))
    if grid_h * grid_w == npos and grid_h == grid_w:
        return pe.contiguous()
    d = pe.shape[-1]
    grid = pe[1:].reshape(1, g0, g0, d).permute(0, 3, 1, 2)
    if mode == "size":
        grid = F.interpolate(grid, size=(grid_h, grid_w), mode="bicubic", align_corners=False)
    elif mode == "scale_factor_0p1":
        grid = F.interpolate(grid, scale_factor=((grid_h + 0.1) / g0, (grid_w + 0.1) / g0), mode="bicubic", align_corners=False)
        if tuple(grid.shape[-2:]) != (grid_h, grid_w):
            raise ValueError("Width or height does not match with the interpolated position embeddings")   # 4.39.3's own check
    else:
        raise ValueError(f"unknown pos_embed_interpolation {mode!r}")
    grid = grid.permute(0, 2, 3, 1).reshape(grid_h * grid_w, d)
    return torch.cat([pe[:1], grid], dim=0).contiguous()


def relative_position_bias(rel_weight: torch.Tensor, seq_len: int, max_distance: int = 128) -> torch.Tensor:
    """MPNetEncoder.compute_position_bias (TF:mpnet/modeling_mpnet.py:312-348) -> (heads, L, L) fp32 CPU.
    Depends only on L, so it is computed once per prompt length on the host."""
    w = rel_weight.detach().float().cpu()
    num_buckets = w.shape[0]
    ctx = torch.arange(seq_len, dtype=torch.long)[:, None]
    mem = torch.arange(seq_len, dtype=torch.long)[None, :]
    n = ctx - mem                                   # = -(memory - context)
    half = num_buckets // 2
    bucket = (n < 0).long() * half
    n = n.abs()
    max_exact = half // 2
    small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)).long()
    large = torch.clamp(large, max=half - 1)
    bucket = bucket + torch.where(small, n, large)
    return w[bucket].permute(2, 0, 1).contiguous()   # (H, L, L)


def _resolve_device(spec) -> torch.device:
    """device / device_map argument -> one indexed cuda device (see RadZeroModel.from_pretrained).  Pure host logic: no GPU call."""
    if spec is None:
        return torch.device("cuda", _current_cuda_index())
    if isinstance(spec, dict):
        targets = {str(_resolve_device(v)) for v in spec.values()}
        if len(targets) != 1:
            raise NotImplementedError(f"device_map places modules on different devices ({sorted(targets)}): one handle runs on one GPU")
        return torch.device(targets.pop())
    if isinstance(spec, int):
        if spec < 0:
            raise ValueError("device index must be >= 0")
        return torch.device("cuda", spec)
    if isinstance(spec, str) and spec in ("auto", "balanced", "balanced_low_0", "sequential"):
        return torch.device("cuda", _current_cuda_index())
    d = torch.device(spec)
    if d.type != "cuda":
        raise RuntimeError(f"RadZeroModel runs on an AMD GPU through libradzero_hip.so; device {spec!r} has no path here")
    return d if d.index is not None else torch.device("cuda", _current_cuda_index())


def _current_cuda_index() -> int:
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


class RadZeroModel:
    """Drop-in for `CxrAlignModel` on the inference path (compute_logits_type == "radzero")."""

    def __init__(self, config: Optional[RadZeroConfig] = None, torch_dtype: torch.dtype = torch.bfloat16,
                 device: str | torch.device = "cuda:0"):
        self.config = config or RadZeroConfig()
        if self.config.sim_op not in ("cos", "dot"):
            raise NotImplementedError(f"sim_op {self.config.sim_op!r}: SimilarityLogit knows 'cos' and 'dot' (losses.py:207-217)")
        if self.config.compute_logits_type not in ("radzero", "cls_alignment", "global_alignment"):
            raise NotImplementedError(f"compute_logits_type {self.config.compute_logits_type!r} (modeling.py:288-353)")
        if self.config.use_text_projection and self.config.compute_logits_type == "radzero":
            # the reference would hand 2 * hidden text features to a VL-CABS head built for hidden (modeling.py:70-73 vs losses.py:35-69)
            raise NotImplementedError("use_text_projection with compute_logits_type 'radzero': the reference's head cannot take projected text features")
        self.dtype = torch_dtype
        if torch.device(device).type != "cuda":
            raise RuntimeError("RadZeroModel runs on an AMD GPU through libradzero_hip.so; there is no CPU path")
        self._device = _resolve_device(device)          # torch.device("cuda") -> the current GPU, with its index
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: radzero_amd has no CPU / PyTorch fallback")
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self._state_dict = None
        self._pos_embed = None
        self._rel_weight = None
        self._text_proj = None                    # (weight (2 D, D), bias (2 D)) fp32 on the device when config.use_text_projection
        self._grids = set()
        self._rel_bias_cache: Dict[int, torch.Tensor] = {}
        self._text_cache: Dict[bytes, torch.Tensor] = {}
        self._text_ident_cache: Dict[tuple, tuple] = {}
        self.text_cache_enabled = True
        self._reserved = (0, 0, 0, 0)
        self._side_stream = None
        self._options: Dict[str, int] = {}        # set through set_model_option: replayed when to() / float() re-creates the handle
        self.training = False
        self._create()

    # ---- lifecycle ---------------------------------------------------------------------------
    def _create(self):
        c = self.config
        rc = _lib.RzConfig(
            compute_dtype=_DTYPES[self.dtype], hidden_size=c.hidden_size, num_attention_heads=c.num_attention_heads,
            mlp_ratio=c.mlp_ratio, patch_size=c.patch_size, num_channels=c.num_channels, vit_layers=c.vit_layers,
            align_layers=c.align_layers, vit_layer_norm_eps=c.vit_layer_norm_eps, vocab_size=c.vocab_size,
            max_position_embeddings=c.max_position_embeddings, text_layers=c.text_layers,
            text_intermediate_size=c.text_intermediate_size, text_layer_norm_eps=c.text_layer_norm_eps,
            pad_token_id=c.pad_token_id, shared_layer_norm_eps=c.shared_layer_norm_eps)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.rz_create(ctypes.byref(rc), ctypes.byref(self._h)), "rz_create")
            _lib.check(self._lib.rz_set_model_option(self._h, b"sim_op", 1 if c.sim_op == "dot" else 0), "rz_set_model_option(sim_op)")
            for name, v in self._options.items():      # the handle's own switches survive a dtype / device move
                _lib.check(self._lib.rz_set_model_option(self._h, name.encode(), v), f"rz_set_model_option({name})")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            with torch.cuda.device(self._device):
                self._lib.rz_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_state_dict(cls, state_dict, config: Optional[RadZeroConfig] = None, torch_dtype=torch.bfloat16,
                        device="cuda:0") -> "RadZeroModel":
        m = cls(config, torch_dtype, device)
        m.load_state_dict(state_dict)
        return m

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=torch.bfloat16, device=None, config: Optional[RadZeroConfig] = None,
                        device_map=None, trust_remote_code: bool = True, **_ignored) -> "RadZeroModel":
        """`AutoModel.from_pretrained("Deepnoid/RadZero", trust_remote_code=True, torch_dtype=..., device_map=...)`
        analogue (README.md:77-82) for a LOCAL checkpoint directory / file (there is no network here).

        `device_map` is honoured the way the README uses it — one device for the whole model: a device string / torch.device /
        GPU index, "auto" / "cuda" / "balanced" / "sequential" (= the current GPU: the model is 0.8 GB, there is nothing to spread),
        or a {"": device} dict.  A map that places modules on DIFFERENT devices, or anything on the CPU / disk, raises: this
        implementation has one handle per GPU and no CPU path."""
        from .checkpoint import config_from_hf, load_checkpoint
        import os
        if device is not None and device_map is not None and _resolve_device(device) != _resolve_device(device_map):
            raise ValueError("from_pretrained: `device` and `device_map` name different devices")
        dev = _resolve_device(device if device is not None else device_map)
        sd = load_checkpoint(path)
        if config is None:
            has_cfg = os.path.isdir(path) and os.path.exists(os.path.join(path, "config.json"))
            config = config_from_hf(path if has_cfg else {}, state_dict=sd)
        return cls.from_state_dict(sd, config, torch_dtype=torch_dtype, device=dev)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts the reference checkpoint's names (numpy arrays or torch tensors, any float dtype)."""
        self._state_dict = state_dict
        proj = {}
        with torch.cuda.device(self._device):
            for name, value in state_dict.items():
                a = value.detach().float().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value, np.float32)
                a = np.ascontiguousarray(a, dtype=np.float32)
                if name == "vision_model.embeddings.position_embeddings":
                    self._pos_embed = torch.from_numpy(a.copy())
                elif name == "text_model.encoder.relative_attention_bias.weight":
                    self._rel_weight = torch.from_numpy(a.copy())
                elif name in ("text_projector.weight", "text_projector.bias"):      # modeling.py:70-73: applied on the Python side of the C-ABI (rz_rows_dot)
                    if not self.config.use_text_projection:
                        raise KeyError(f"{name} in the checkpoint but config.use_text_projection is False")
                    proj[name] = torch.from_numpy(a.copy()).to(self._device)
                    continue
                _lib.check(self._lib.rz_load_weight(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                           f"rz_load_weight({name})")
            if strict:
                _lib.check(self._lib.rz_weights_ready(self._h), "rz_weights_ready")
                if self._pos_embed is None or self._rel_weight is None:
                    raise _lib.RzError("checkpoint lacks position_embeddings / relative_attention_bias")
                if self.config.use_text_projection and len(proj) != 2:
                    raise _lib.RzError("config.use_text_projection but the checkpoint lacks text_projector.weight / .bias")
        if proj:
            self._text_proj = (proj["text_projector.weight"].contiguous(), proj["text_projector.bias"].contiguous())
        self._grids.clear()
        self._rel_bias_cache.clear()
        self._text_cache.clear()
        self._text_ident_cache.clear()
        return self

    # ---- nn.Module-like surface used by the reference's callers -------------------------------
    @property
    def device(self) -> torch.device:
        return self._device

    def eval(self):
        self.training = False
        return self

    def to(self, *args, **kwargs):
        """`model.to(torch.float32)` (inference/utils.py:37) re-packs the weights in the new compute dtype."""
        dtype = kwargs.get("dtype")
        device = kwargs.get("device")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, torch.device)):
                device = a
        if device is not None and torch.device(device).type != "cuda":
            raise RuntimeError("RadZeroModel cannot be moved off the GPU (no CPU path)")
        new_device = _resolve_device(device) if device is not None else self._device
        if (dtype is not None and dtype != self.dtype) or new_device != self._device:
            if dtype is not None and dtype not in _DTYPES:
                raise NotImplementedError(f"unsupported dtype {dtype}")
            sd = self._state_dict
            self.close()
            self.dtype = dtype or self.dtype
            self._device = new_device
            self._reserved = (0, 0, 0, 0)
            self._create()
            if sd is not None:
                self.load_state_dict(sd)
        return self

    def float(self):
        return self.to(torch.float32)

    # ---- internals -----------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _ensure(self, batch=0, tokens=0, prompts=0, plen=0):
        r = self._reserved
        if batch > r[0] or tokens > r[1] or prompts > r[2] or plen > r[3]:
            new = (max(batch, r[0]), max(tokens, r[1]), max(prompts, r[2]), max(plen, r[3]))
            torch.cuda.synchronize(self._device)
            _lib.check(self._lib.rz_reserve(self._h, *new), "rz_reserve")
            self._reserved = new

    def _ensure_grid(self, gh, gw):
        if (gh, gw) not in self._grids:
            if self._pos_embed is None:
                raise _lib.RzError("weights not loaded")
            table = interpolate_pos_encoding(self._pos_embed, gh, gw, self.config.pos_embed_interpolation).numpy()
            torch.cuda.synchronize(self._device)
            _lib.check(self._lib.rz_set_position_table(self._h, gh, gw, table.ctypes.data_as(ctypes.c_void_p)),
                       "rz_set_position_table")
            self._grids.add((gh, gw))

    def _vision(self, pixel_values: torch.Tensor, want_tokens: bool):
        if pixel_values.dim() != 4:
            raise ValueError("pixel_values must be (batch, channels, height, width)")
        px = pixel_values.to(device=self._device, dtype=torch.float32).contiguous()
        b, c, hh, ww = px.shape
        p = self.config.patch_size
        gh, gw = hh // p, ww // p
        n = gh * gw + 1
        with torch.cuda.device(self._device):
            if c == self.config.num_channels and gh > 0 and gw > 0 and b > 0:
                self._ensure_grid(gh, gw)
                self._ensure(batch=b, tokens=n)
            tokens = torch.empty((b, n, self.config.hidden_size), dtype=torch.float32, device=self._device) if want_tokens else None
            _lib.check(self._lib.rz_vision_forward(self._h, _ptr(px), b, c, hh, ww, _ptr(tokens), self._stream()),
                       "rz_vision_forward")
        return tokens, (b, n)

    # ---- CxrAlignModel.forward_vision_model (modeling.py:96-123) -------------------------------
    @torch.no_grad()
    def forward_vision_model(self, pixel_values):
        tokens, _ = self._vision(pixel_values, want_tokens=True)
        cls_token = tokens[:, 0]
        patch_tokens = tokens[:, 1:]
        b, n, d = tokens.shape
        with torch.cuda.device(self._device):
            image_features = torch.empty((b, 2 * d), dtype=torch.float32, device=self._device)
            _lib.check(self._lib.rz_image_features(_ptr(tokens), n, b, n, d, _ptr(image_features), self._stream()), "rz_image_features")
        return {"vision_tokens": tokens, "image_cls_token": cls_token, "image_patch_tokens": patch_tokens,
                "image_features": image_features}

    # ---- CxrAlignModel.forward_text_model, MPNet branch (modeling.py:125-211) ------------------
    def _check_ids(self, lo: int, hi: int):
        if lo < 0 or hi >= self.config.vocab_size:
            raise IndexError("index out of range in self")          # nn.Embedding's error for bad token ids

    def _text_forward_raw(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """rz_text_forward on the CURRENT stream for validated int64 device tensors (n_prompts, len); no cache, no host sync."""
        t, l = ids.shape
        with torch.cuda.device(self._device):
            if l not in self._rel_bias_cache:
                self._rel_bias_cache[l] = relative_position_bias(self._rel_weight, l).to(self._device)
            self._ensure(prompts=t, plen=l)
            feat = torch.empty((t, self.config.hidden_size), dtype=torch.float32, device=self._device)
            _lib.check(self._lib.rz_text_forward(self._h, _ptr(ids), _ptr(mask), t, l, _ptr(self._rel_bias_cache[l]),
                                                 _ptr(feat), self._stream()), "rz_text_forward")
        return feat

    @torch.no_grad()
    def forward_text_model(self, encoded_input):
        ids = encoded_input["input_ids"].to(device=self._device, dtype=torch.int64).contiguous()
        mask = encoded_input["attention_mask"].to(device=self._device, dtype=torch.int64).contiguous()
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError("input_ids / attention_mask must be (n_prompts, len) and agree")
        if not torch.cuda.is_current_stream_capturing():
            lo, hi = torch.stack([ids.min(), ids.max()]).tolist()      # ONE device->host copy for both bounds
            self._check_ids(int(lo), int(hi))
        feat = self._project_text(self._text_forward_raw(ids, mask))
        return {"text_features_wo_l2_norm": feat, "text_features": F.normalize(feat, p=2, dim=1)}

    def _rows_dot(self, a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  rows_per_group: int = 0, strides=None) -> torch.Tensor:
        """out[m][n] = a[m] . b[n] (+ bias[n]) in fp32 through rz_rows_dot; `strides` = (group, row, column) strides of `out` for a transposed result."""
        m, k = a.shape
        nb = b.shape[0]
        if a.dtype != torch.float32 or b.dtype != torch.float32 or a.stride(1) != 1 or b.stride(1) != 1 or b.shape[1] < k or m == 0:
            raise ValueError("_rows_dot: fp32 operands with contiguous rows, b at least as wide as a")
        if a.shape[0] == 1:                  # a one-row view reports an arbitrary row stride
            a = a.reshape(1, k).contiguous()
        with torch.cuda.device(self._device):
            if out is None:
                out = torch.empty((m, nb), dtype=torch.float32, device=self._device)
                rows_per_group, strides = m, (0, nb, 1)
            _lib.check(self._lib.rz_rows_dot(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(bias), _ptr(out), m, nb, k, rows_per_group,
                                             strides[0], strides[1], strides[2], self._stream()), "rz_rows_dot")
        return out

    def _project_text(self, feat: torch.Tensor) -> torch.Tensor:
        """text_projector (modeling.py:199-200) when the configuration has one."""
        if self._text_proj is None:
            return feat
        return self._rows_dot(feat.contiguous(), self._text_proj[0], self._text_proj[1])

    def _cache_put(self, key: bytes, feat: torch.Tensor):
        self._text_cache[key] = feat
        while len(self._text_cache) > 512:                           # bounded: a service that sees a new text per request must not grow
            self._text_cache.pop(next(iter(self._text_cache)))

    def _ident(self, ids, mask):
        # inference tensors (created under torch.inference_mode()) keep no version counter — reading `_version` raises — so an
        # in-place write to them could not be noticed: they take the content key
        if torch.is_tensor(ids) and torch.is_tensor(mask) and not ids.is_inference() and not mask.is_inference():
            return (ids.data_ptr(), mask.data_ptr(), tuple(ids.shape), tuple(mask.shape), ids.dtype, mask.dtype,
                    str(ids.device), ids._version, mask._version)
        return None

    def _ident_put(self, ident, ids, mask, feat):
        if ident is not None:
            if len(self._text_ident_cache) >= 64:                   # bounded: drop the oldest identity entry
                self._text_ident_cache.pop(next(iter(self._text_ident_cache)))
            self._text_ident_cache[ident] = (ids, mask, feat)

    def encode_prompts(self, encoded) -> torch.Tensor:
        """text_features_wo_l2_norm for a prompt set, cached: the reference re-encodes every prompt for every
        image batch (modeling.py:290-307); the embeddings do not depend on the images.

        Two levels.  (1) identity: (data_ptr, shape, dtype, device, _version) of the id / mask tensors — no device
        work, no host sync; this is what every reference caller hits, because they tokenise once and pass the same
        tensors for every image batch (inference/utils.py:92-100, grounding_utils.py:50-62).  The entry keeps the two
        tensors alive, so their storage cannot be recycled under the key, and an in-place write bumps `_version`.
        (2) content: the token bytes, for callers that re-tokenise per call (one D2H copy of a few hundred bytes, which also
        carries the token-id range check); at most 512 prompt sets are kept."""
        ids = encoded["input_ids"]
        mask = encoded["attention_mask"]
        if not self.text_cache_enabled:
            return self.forward_text_model({"input_ids": ids, "attention_mask": mask})["text_features_wo_l2_norm"]
        ident = self._ident(ids, mask)
        if ident is not None:
            hit = self._text_ident_cache.get(ident)
            if hit is not None:
                return hit[2]
        feat = self._encode_by_content(ids, mask)
        self._ident_put(ident, ids, mask, feat)
        return feat

    def _encode_by_content(self, ids, mask) -> torch.Tensor:
        """Content-keyed lookup / encode on the CURRENT stream: one host copy of the tokens serves the key and the id range check."""
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError("input_ids / attention_mask must be (n_prompts, len) and agree")
        hi_, hm_ = ids.detach().to("cpu", torch.int64), mask.detach().to("cpu", torch.int64)
        key = hi_.numpy().tobytes() + b"|" + hm_.numpy().tobytes() + str(tuple(ids.shape)).encode()
        feat = self._text_cache.get(key)
        if feat is None:
            self._check_ids(int(hi_.min()), int(hi_.max()))
            # projected like forward_text_model's output (text_projector, modeling.py:199-200): cached and uncached paths hand out the same features
            feat = self._project_text(self._text_forward_raw(ids.to(device=self._device, dtype=torch.int64).contiguous(),
                                                             mask.to(device=self._device, dtype=torch.int64).contiguous()))
            self._cache_put(key, feat)
        return feat

    def _encode_beside_vision(self, encoded, ids_ready: "torch.cuda.Event") -> torch.Tensor:
        """The per-request path (a NEW text with every image: eval_refer_grounding, grounding_utils.py:283-326; extract_similarity_map,
        attention_map_base.py:12-42): the text encoder runs on a side stream BESIDE the vision forward that the caller has already
        enqueued on the current stream (disjoint workspaces inside the handle; the two only meet in rz_vlcabs), and the host's one
        blocking copy of the tokens waits for `ids_ready` on that side stream, not for the vision forward.  The current stream is
        made to wait for the embeddings before this returns.  Under stream capture the same fork / join is recorded (no cache)."""
        ids, mask = encoded["input_ids"], encoded["attention_mask"]
        main = torch.cuda.current_stream(self._device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self._device)
        side = self._side_stream
        capturing = torch.cuda.is_current_stream_capturing()
        with torch.cuda.stream(side):
            side.wait_event(ids_ready)
            if capturing or not self.text_cache_enabled:
                if ids.dim() != 2 or ids.shape != mask.shape:
                    raise ValueError("input_ids / attention_mask must be (n_prompts, len) and agree")
                dev_ids = ids.to(device=self._device, dtype=torch.int64).contiguous()
                if not capturing:           # the reference's IndexError for ids outside the vocabulary (one host read of both bounds, as forward_text_model)
                    lo, hi = torch.stack([dev_ids.min(), dev_ids.max()]).tolist()
                    self._check_ids(int(lo), int(hi))
                feat = self._project_text(self._text_forward_raw(dev_ids, mask.to(device=self._device, dtype=torch.int64).contiguous()))
            else:
                ident = self._ident(ids, mask)
                feat = self._encode_by_content(ids, mask)
                self._ident_put(ident, ids, mask, feat)
            done = torch.cuda.Event()
            done.record(side)
        main.wait_event(done)
        feat.record_stream(main)
        return feat

    # ---- CxrAlignModel.compute_logits, compute_logits_type == "radzero" (modeling.py:278-328) ---
    @torch.no_grad()
    def compute_logits(self, pixel_values, encoded_key_phrases, text_features: Optional[torch.Tensor] = None, **kwargs):
        """kwargs (encoded_negative_phrases, use_negative_logits, ...) are accepted and ignored, as in the
        reference (modeling.py:282).  `text_features` lets a data-parallel driver pass all-gathered embeddings."""
        if self.config.compute_logits_type != "radzero":
            return self._compute_alignment_logits(pixel_values, encoded_key_phrases, text_features)
        pending = None
        if text_features is None:
            enc = encoded_key_phrases[0]
            ident = self._ident(enc["input_ids"], enc["attention_mask"]) if self.text_cache_enabled else None
            hit = self._text_ident_cache.get(ident) if ident is not None else None
            if hit is not None:
                text_features = hit[2]          # the same tokenised tensors as in an earlier call: every batch loop of the reference
            else:
                # a text this model has not been handed before: encode it BESIDE the vision forward (see _encode_beside_vision)
                with torch.cuda.device(self._device):
                    pending = torch.cuda.Event()
                    pending.record(torch.cuda.current_stream(self._device))
        _, (b, n) = self._vision(pixel_values, want_tokens=False)
        if pending is not None:
            text_features = self._encode_beside_vision(enc, pending)
        text_features = text_features.to(device=self._device, dtype=torch.float32).contiguous()
        t = text_features.shape[0]
        with torch.cuda.device(self._device):
            self._ensure(batch=b, tokens=n, prompts=t)
            scores = torch.empty((b, t, n), dtype=torch.float32, device=self._device)
            t2i = torch.empty((t, b), dtype=torch.float32, device=self._device)
            logits = torch.empty((b, t), dtype=torch.float32, device=self._device)
            _lib.check(self._lib.rz_vlcabs(self._h, _ptr(text_features), t, b, _ptr(scores), _ptr(t2i), _ptr(logits),
                                           self._stream()), "rz_vlcabs")
        outputs = {"t2i_attn_weights": [scores]}
        # mean over the (single-element) list, then drop the CLS column (modeling.py:311-317)
        outputs["similarity_scores"] = scores[:, :, 1:] if self.config.use_vision_cls_token else scores
        # the reference's `.squeeze()` in SimilarityLogit (losses.py:229-233) collapses B==1 / T==1:
        t2i_sq = t2i.squeeze()
        outputs["t2i_logits"] = t2i_sq
        lg = logits.squeeze()
        outputs["logits"] = lg if lg.dim() > 0 else lg.reshape(1)     # / exp((1,)-param) makes a 0-d result (1,)
        return outputs

    # ---- compute_logits_type "cls_alignment" / "global_alignment" (modeling.py:330-353) --------
    def _compute_alignment_logits(self, pixel_values, encoded_key_phrases, text_features: Optional[torch.Tensor] = None):
        """`text_features`: the prompt groups' un-normalised features (encode_prompts / a data-parallel driver's all-gathered table), concatenated
        in group order; otherwise every group is encoded through the text cache (the reference re-encodes them for every image batch)."""
        tokens, (b, n) = self._vision(pixel_values, want_tokens=True)
        if text_features is None:
            text_features = torch.cat([self.encode_prompts(kp) for kp in encoded_key_phrases], dim=0)
        key_features = F.normalize(text_features.to(device=self._device, dtype=torch.float32), p=2, dim=1).contiguous()     # (N_total, D') "text_features"
        t, d = key_features.shape[0], self.config.hidden_size
        outputs = {}
        if self.config.compute_logits_type == "cls_alignment":
            if key_features.shape[1] != d:
                raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({b}x{d} and {key_features.shape[1]}x{t})")     # torch's error in the reference
            outputs["logits"] = self._rows_dot(tokens[:, 0], key_features)                      # image_cls_token @ key_features.T
            return outputs
        if key_features.shape[1] != 2 * d:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({b}x{2 * d} and {key_features.shape[1]}x{t})")
        with torch.cuda.device(self._device):
            image_features = torch.empty((b, 2 * d), dtype=torch.float32, device=self._device)
            _lib.check(self._lib.rz_image_features(_ptr(tokens), n, b, n, d, _ptr(image_features), self._stream()), "rz_image_features")
            outputs["logits"] = self._rows_dot(image_features, key_features)                    # image_features @ key_features.T
            # einsum("ind,jd->ijn", image_patch_tokens, key_features[:, hidden:]) -> (B, N_total, L): rows = every token of every image (the cls
            # row lands in column 0 of a (B, T, 1 + L) buffer and is sliced off: patch rows are not contiguous across images)
            sim = torch.empty((b, t, n), dtype=torch.float32, device=self._device)
            self._rows_dot(tokens.reshape(b * n, d), key_features[:, d:], out=sim, rows_per_group=n, strides=(t * n, 1, n))
        outputs["similarity_scores"] = sim[:, :, 1:]
        return outputs

    # ---- similarity-map post-processing (segmentation_utils.py:62-70, attention_map_base.py:57) ---
    @torch.no_grad()
    def upsample_similarity(self, similarity_scores: torch.Tensor, size, sigmoid: bool = False,
                            keep_aspect_ratio: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(..., g*g) patch-grid scores -> (..., H, W) bilinear (align_corners=False) [+ sigmoid].
        keep_aspect_ratio: the AspectRatioBlipImageProcessor branch (upsample to the padded square, crop the image area).
        out: optional contiguous fp32 buffer of n_maps*H*W elements to write into (a serving loop re-uses one 4.3 GB buffer at
        BASELINE cfg 4 instead of allocating it per batch)."""
        hh, ww = int(size[0]), int(size[1])
        g = int(round(math.sqrt(similarity_scores.shape[-1])))
        if g * g != similarity_scores.shape[-1]:
            raise ValueError("last dim must be a square patch grid")
        lead = similarity_scores.shape[:-1]
        s = similarity_scores.to(device=self._device, dtype=torch.float32)
        n_maps = int(np.prod(lead)) if len(lead) else 1
        if s.dim() == 3 and s.stride(2) == 1 and s.stride(0) == s.shape[1] * s.stride(1):
            flat, stride = s, s.stride(1)          # e.g. the [:, :, 1:] view of the (B, T, N) score tensor: no copy
        else:
            flat, stride = s.contiguous(), g * g
        if out is None:
            out = torch.empty((n_maps, hh, ww), dtype=torch.float32, device=self._device)
        elif out.numel() != n_maps * hh * ww or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self._device:
            raise ValueError("out must be a contiguous fp32 tensor of n_maps*H*W elements on the model's device")
        with torch.cuda.device(self._device):
            _lib.check(self._lib.rz_upsample_maps_ex(self._h, _ptr(flat), stride, n_maps, g, hh, ww, int(sigmoid),
                                                     int(keep_aspect_ratio), _ptr(out), self._stream()), "rz_upsample_maps")
        return out.reshape(*lead, hh, ww)

    @torch.no_grad()
    def grounding_points(self, similarity_scores: torch.Tensor, size, keep_aspect_ratio: bool = False) -> torch.Tensor:
        """get_grounding_point (grounding_utils.py:166-261) for every map: (..., g*g) -> (..., 2) int32 (x, y) of the
        first maximum of the bilinear-upsampled map; the (H, W) map itself is never materialised."""
        hh, ww = int(size[0]), int(size[1])
        g = int(round(math.sqrt(similarity_scores.shape[-1])))
        if g * g != similarity_scores.shape[-1]:
            raise ValueError("last dim must be a square patch grid")
        lead = similarity_scores.shape[:-1]
        s = similarity_scores.to(device=self._device, dtype=torch.float32)
        n_maps = int(np.prod(lead)) if len(lead) else 1
        if s.dim() == 3 and s.stride(2) == 1 and s.stride(0) == s.shape[1] * s.stride(1):
            flat, stride = s, s.stride(1)
        else:
            flat, stride = s.contiguous(), g * g
        xy = torch.empty((n_maps, 2), dtype=torch.int32, device=self._device)
        ws = torch.empty((n_maps,), dtype=torch.int64, device=self._device)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.rz_grounding_points_ex(self._h, _ptr(flat), stride, n_maps, g, hh, ww, int(keep_aspect_ratio),
                                                        _ptr(xy), _ptr(ws), self._stream()), "rz_grounding_points")
        return xy.reshape(*lead, 2)

    # ---- hipGraph replay for latency-bound (single / small image) calls --------------------------
    def make_graphed(self, pixel_shape, encoded_key_phrases, maps: str = "none"):
        """Capture compute_logits for a fixed pixel shape + prompt set into a HIP graph (≈140 kernel launches per call
        collapse into one replay).  Returns `run(pixel_values) -> outputs`; the returned tensors are the graph's static
        buffers and are overwritten by the next `run`."""
        enc = {k: v.to(self._device) for k, v in encoded_key_phrases[0].items()}
        feats = self.encode_prompts(enc)
        static_px = torch.zeros(tuple(pixel_shape), dtype=torch.float32, device=self._device)
        with torch.cuda.device(self._device):
            side = torch.cuda.Stream(device=self._device)
            side.wait_stream(torch.cuda.current_stream(self._device))
            with torch.cuda.stream(side):                       # warm-up off the capture path: workspaces, tables
                for _ in range(2):
                    self.compute_logits(static_px, [enc], text_features=feats)
            torch.cuda.current_stream(self._device).wait_stream(side)
            torch.cuda.synchronize(self._device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.compute_logits(static_px, [enc], text_features=feats)
                if maps == "points":
                    out["grounding_points"] = self.grounding_points(out["similarity_scores"], pixel_shape[-2:])

        def run(pixel_values: torch.Tensor):
            static_px.copy_(pixel_values.to(device=self._device, dtype=torch.float32), non_blocking=True)
            graph.replay()
            return out

        run.graph = graph
        return run

    def make_graphed_request(self, pixel_shape, prompt_shape, points: bool = True):
        """The per-request path (one image + one NEW text per call: eval_refer_grounding, grounding_utils.py:283-326) as ONE HIP graph:
        text encoder (on a forked branch) + vision forward + VL-CABS [+ fused grounding point].  Returns
        run(pixel_values, input_ids, attention_mask) -> outputs (the graph's static buffers, overwritten by the next run).  Token ids are
        NOT range-checked here (no host round trip inside a replay): validate them where they are produced."""
        t, l = int(prompt_shape[0]), int(prompt_shape[1])
        static_px = torch.zeros(tuple(pixel_shape), dtype=torch.float32, device=self._device)
        static_ids = torch.full((t, l), self.config.pad_token_id, dtype=torch.int64, device=self._device)
        static_ids[:, 0] = 0
        static_mask = torch.ones((t, l), dtype=torch.int64, device=self._device)
        enc = {"input_ids": static_ids, "attention_mask": static_mask}
        cache_was = self.text_cache_enabled
        self.text_cache_enabled = False
        try:
            with torch.cuda.device(self._device):
                warm = torch.cuda.Stream(device=self._device)
                warm.wait_stream(torch.cuda.current_stream(self._device))
                with torch.cuda.stream(warm):                   # warm-up off the capture path: workspaces, tables, the side stream
                    for _ in range(2):
                        o = self.compute_logits(static_px, [enc])
                        if points:
                            self.grounding_points(o["similarity_scores"], pixel_shape[-2:])
                torch.cuda.current_stream(self._device).wait_stream(warm)
                torch.cuda.synchronize(self._device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self.compute_logits(static_px, [enc])
                    if points:
                        out["grounding_points"] = self.grounding_points(out["similarity_scores"], pixel_shape[-2:])
        finally:
            self.text_cache_enabled = cache_was

        def run(pixel_values: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor):
            static_px.copy_(pixel_values.to(device=self._device, dtype=torch.float32), non_blocking=True)
            static_ids.copy_(input_ids.to(device=self._device, dtype=torch.int64), non_blocking=True)
            static_mask.copy_(attention_mask.to(device=self._device, dtype=torch.int64), non_blocking=True)
            graph.replay()
            return out

        run.graph = graph
        return run

    # ---- library switches ------------------------------------------------------------------------
    @staticmethod
    def set_option(name: str, value: int) -> None:
        """PROCESS-WIDE switch of the HIP library (include/radzero_hip.h, rz_set_option): for A/B tools.  Handles that have not set
        the option themselves follow it."""
        lib = _lib.load()
        _lib.check(lib.rz_set_option(name.encode(), int(value)), "rz_set_option")

    def set_model_option(self, name: str, value: Optional[int]) -> None:
        """This model's own switch (rz_set_model_option), e.g. set_model_option("gemm_f32_split", 0): fp32 mode on the exact-fp32
        MFMA kernels instead of the hi/lo-split f16 ones — other models of the process are unaffected.  None = follow the
        process-wide value again."""
        v = -(2 ** 31) if value is None else int(value)
        with torch.cuda.device(self._device):
            _lib.check(self._lib.rz_set_model_option(self._h, name.encode(), v), "rz_set_model_option")
        if value is None:
            self._options.pop(name, None)
        else:
            self._options[name] = v
        if name == "pad_rows":                       # the library dropped its tables / workspace sizes
            self._grids.clear()
            self._reserved = (0, 0, 0, 0)

    F32_PRECISIONS = {"high": {"attn_f32_pv": 0}, "fast": {"attn_f32_pv": 1}}

    def set_f32_precision(self, level: str) -> None:
        """fp32 (1e-3) mode only: which correction terms the split products keep (include/radzero_hip.h "attn_f32_pv";
        profiles/r05/fp32_term_ablation.log).  "high" (default): every product at 22 bits — 7e-5 from the reference on the goldens,
        3e-5 on the outlier-channel checkpoint.  "fast": the attention's P V product on the f16 hi planes alone — 3.1e-4 on the goldens,
        but 1.2e-3 on the outlier-channel checkpoint (outside the 1e-3 contract there), 19 % more images per second."""
        if level not in self.F32_PRECISIONS:
            raise ValueError(f"f32_precision {level!r}: 'high' or 'fast'")
        for k, v in self.F32_PRECISIONS[level].items():
            self.set_model_option(k, v)

    # ---- batch shaping (host logic) ---------------------------------------------------------------
    def padded_tokens(self, n_tokens: int, batch: int = 0) -> int:
        """Token rows per image a batch of `batch` images gets (rz_padded_tokens: the handle's pad_rows rule)."""
        v = ctypes.c_int(0)
        _lib.check(self._lib.rz_padded_tokens(self._h, int(n_tokens), int(batch), ctypes.byref(v)), "rz_padded_tokens")
        return int(v.value)

    def preferred_batch(self, batch: int, height: int, width: int, max_drop: Optional[int] = None, min_gain: float = 0.02) -> int:
        """The forward size <= `batch` a batch driver should use when images of this size arrive in batches of `batch`
        (radzero_amd/shaping.py: whole rounds of the persistent GEMM's 256 tiles; 518^2 x 64 -> 62)."""
        from .shaping import preferred_batch
        p = self.config.patch_size
        return preferred_batch(int(batch), (height // p) * (width // p) + 1, self.padded_tokens, self.config.hidden_size, self.config.mlp_ratio, max_drop, min_gain)

    def gemm_tile_cost(self, batch: int, n_tokens: int) -> float:
        """radzero_amd/shaping.py gemm_tile_cost under this handle's padding rule (diagnostics: tools/shaping_sweep.py)."""
        from .shaping import gemm_tile_cost
        return gemm_tile_cost(int(batch), int(n_tokens), self.padded_tokens, self.config.hidden_size, self.config.mlp_ratio)

    def guard_reruns(self) -> int:
        """fp32 mode: forwards repeated on the exact-fp32 kernels because a value left the f16 planes' range (a checkpoint that trips the
        guard on every forward runs at a quarter of the speed: the batch driver warns once, bench.py prints the count)."""
        return self.get_model_option("f32_split_guard_reruns")

    def get_model_option(self, name: str) -> int:
        v = ctypes.c_int(0)
        _lib.check(self._lib.rz_get_model_option(self._h, name.encode(), ctypes.byref(v)), "rz_get_model_option")
        return int(v.value)

    # ---- measurement -----------------------------------------------------------------------------
    def profile(self, enable: bool, families=None):
        """HIP-event timing of kernel families on the launch stream; `families` (e.g. ("attn",)) restricts the recording to those."""
        code = int(bool(enable))
        if enable and families is not None:
            code = 1 + (sum(1 << _lib.PROF_FAMILIES.index(f) for f in families) << 1)
        _lib.check(self._lib.rz_profile_enable(self._h, code), "rz_profile_enable")

    def profile_read(self):
        ms = (ctypes.c_float * len(_lib.PROF_FAMILIES))()
        n = (ctypes.c_int64 * len(_lib.PROF_FAMILIES))()
        _lib.check(self._lib.rz_profile_read(self._h, ms, n), "rz_profile_read")
        return {f: {"ms": float(ms[i]), "launches": int(n[i])} for i, f in enumerate(_lib.PROF_FAMILIES)}
