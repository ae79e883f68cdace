"""CPU-only: libradzero_hip.so builds/loads and exports every symbol include/radzero_hip.h declares.
No compute call is made (there is no GPU here) — but the no-fallback contract is checked."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "radzero_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rz_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    from radzero_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/radzero_hip.h but not exported"
    assert set(declared) == set(_lib.SYMBOLS), "ctypes prototype table out of sync with the header"
    assert lib.rz_version().decode().startswith("radzero_hip")


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radzero_amd.modeling import RadZeroModel
    with pytest.raises(RuntimeError):
        RadZeroModel(device="cuda:0")
    with pytest.raises(RuntimeError):
        RadZeroModel(device="cpu")
    # the C-ABI itself refuses to create a handle without a device
    from radzero_amd import _lib
    lib = _lib.load()
    cfg = _lib.RzConfig(compute_dtype=1, hidden_size=768, num_attention_heads=12, mlp_ratio=4, patch_size=14,
                        num_channels=3, vit_layers=12, align_layers=2, vit_layer_norm_eps=1e-6, vocab_size=30527,
                        max_position_embeddings=514, text_layers=12, text_intermediate_size=3072,
                        text_layer_norm_eps=1e-5, pad_token_id=1, shared_layer_norm_eps=1e-5)
    h = ctypes.c_void_p()
    assert lib.rz_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"no HIP device" in lib.rz_last_error()


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle (or /root/reference)."""
    pkg = os.path.join(ROOT, "radzero_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "/root/reference" not in src, f


def test_host_tables_match_oracle():
    """Host-side one-time tables (pos-embed bicubic resize, MPNet relative bias) vs the oracle's restatement."""
    import numpy as np
    import torch
    from oracle.radzero_oracle import interpolate_pos_encoding as o_interp, relative_position_bucket_table
    from radzero_amd.modeling import interpolate_pos_encoding, relative_position_bias
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(1, 257, 768, generator=g)
    for gh in (16, 19, 37, 73):
        a = interpolate_pos_encoding(pe, gh, gh)
        b = o_interp(pe, gh, gh)[0]
        assert a.shape == (1 + gh * gh, 768) and torch.equal(a, b)
    w = torch.randn(32, 12, generator=g)
    for L in (1, 7, 32, 130):
        bias = relative_position_bias(w, L)
        tbl = relative_position_bucket_table(L)
        assert torch.equal(bias, w[tbl].permute(2, 0, 1))
        assert np.all(np.isfinite(bias.numpy()))


def test_set_option_known_and_unknown_names():
    """RadZeroModel.set_option -> rz_set_option: every documented switch is accepted (and restored), an unknown name raises ValueError.
    No compute call: runs without a GPU."""
    from radzero_amd.modeling import RadZeroModel
    defaults = {"gemm_variant": 0, "attn_variant": 0, "ln_fused": 1, "attn_f32_split": 1, "gemm_f32_split": 1,
                "pad_rows": 0, "f32_split_guard": 1, "gemm_v1_only": 0, "sim_op": 0, "gemm_f32_mx": 1, "attn_f32_mx": 1,
                "attn_f32_pv": 0, "f32_drop": 0, "gemm_small_tile": 0, "gemm_qkv_pair": 1}
    for name, value in defaults.items():
        RadZeroModel.set_option(name, value)
    with pytest.raises(ValueError):
        RadZeroModel.set_option("no_such_option", 1)
    for retired in ("gemm_skew", "vision_chunk", "vision_streams", "mlp_chunk", "gemm_raster"):      # experiments: not in the product library
        with pytest.raises(ValueError):
            RadZeroModel.set_option(retired, 0)
    # the header documents exactly the options the library knows (gemm_f32_split was missing, gemm_skew stale)
    hdr = open(os.path.join(ROOT, "include", "radzero_hip.h")).read()
    documented = set(re.findall(r'^ \*\s+"([a-z_0-9]+)"', hdr, flags=re.M))
    assert documented == set(defaults), (documented ^ set(defaults))


def test_model_options_need_a_handle():
    """rz_set_model_option / rz_get_model_option reject a null handle (no GPU needed); with a handle: tests/test_gpu_boundary.py."""
    from radzero_amd import _lib
    lib = _lib.load()
    v = ctypes.c_int(0)
    assert lib.rz_set_model_option(None, b"ln_fused", 0) == 10001
    assert lib.rz_get_model_option(None, b"ln_fused", ctypes.byref(v)) == 10001
    assert lib.rz_debug_buffer(b"gemm_v8_stamps", None) == 10001          # tools build only


def test_device_map_resolution():
    """README.md:77-82 passes device_map= to from_pretrained: one device for the whole model is honoured, anything else raises."""
    import torch
    from radzero_amd.modeling import _resolve_device
    assert _resolve_device("cuda:3") == torch.device("cuda", 3)
    assert _resolve_device(2) == torch.device("cuda", 2)
    assert _resolve_device({"": "cuda:1"}) == torch.device("cuda", 1)
    assert _resolve_device({"vision_model": 1, "text_model": "cuda:1"}) == torch.device("cuda", 1)
    for spec in ("cuda", "auto", None, torch.device("cuda")):
        d = _resolve_device(spec)
        assert d.type == "cuda" and d.index is not None          # never an index-less device: the handle is bound to one GPU
    with pytest.raises(NotImplementedError):
        _resolve_device({"vision_model": 0, "text_model": 1})
    for bad in ("cpu", {"": "cpu"}, {"": "disk"}):
        with pytest.raises(RuntimeError):
            _resolve_device(bad)


def test_encode_prompts_cache_under_inference_mode():
    """inference tensors keep no version counter; the identity cache level must step aside for them instead of raising.
    Host logic only: the text encoder is replaced by a counter."""
    import torch
    from radzero_amd.modeling import RadZeroModel
    m = RadZeroModel.__new__(RadZeroModel)
    from radzero_amd.config import RadZeroConfig
    m.config = RadZeroConfig()            # token ids are range-checked before the content key is built
    m.text_cache_enabled = True
    m._text_cache, m._text_ident_cache = {}, {}
    m._text_proj = None                   # no text projector (the released configuration)
    calls = []

    def fake_text(ids, mask):            # stands in for rz_text_forward (the content-keyed level calls the raw encoder)
        calls.append(1)
        return ids.float().sum(1, keepdim=True)
    m._text_forward_raw = fake_text
    m._device = torch.device("cpu")
    with torch.inference_mode():
        ids = torch.arange(12).reshape(3, 4).clone()
        mask = torch.ones_like(ids)
        assert ids.is_inference()
        a = m.encode_prompts({"input_ids": ids, "attention_mask": mask})
        b = m.encode_prompts({"input_ids": ids, "attention_mask": mask})          # content key hit
        assert len(calls) == 1 and torch.equal(a, b) and not m._text_ident_cache
        ids2 = ids + 1
        m.encode_prompts({"input_ids": ids2, "attention_mask": mask})
        assert len(calls) == 2
    ids3 = torch.arange(12).reshape(3, 4) + 7                                      # ordinary tensors: identity level, and it sees in-place writes
    mask3 = torch.ones_like(ids3)
    m.encode_prompts({"input_ids": ids3, "attention_mask": mask3})
    m.encode_prompts({"input_ids": ids3, "attention_mask": mask3})
    assert len(calls) == 3 and len(m._text_ident_cache) == 1
    ids3.add_(1)
    m.encode_prompts({"input_ids": ids3, "attention_mask": mask3})
    assert len(calls) == 4
    m._h = None
