"""GPU tests of the rows either side of compute_logits against fixtures produced by the REFERENCE's own functions
(tools/make_goldens_post.py): map post-processing for both image-processor branches, grounding points, the HF checkpoint
layout through `RadZeroModel.from_pretrained`, the processor-dependent `extract_similarity_map`, and the data-parallel
driver over a real RCCL process group (one rank, fresh child process)."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden, post_map_cases
from radzero_amd.synthetic import synthetic_pixels, synthetic_prompts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_f32(cfg, state_dict):
    from radzero_amd.modeling import RadZeroModel
    m = RadZeroModel.from_state_dict(state_dict, cfg, torch_dtype=torch.float32, device="cuda:0").eval()
    yield m
    m.close()


def test_upsample_and_grounding_match_reference_functions(model_f32):
    """rz_upsample_maps_ex / rz_grounding_points_ex vs interpolate_similarity_scores (segmentation_utils.py:36-70) and
    get_grounding_point (grounding_utils.py:166-261) run by the reference itself, Blip and AspectRatioBlip branches,
    square and non-square originals up to 1200x900.  Maps: fp32 interpolation arithmetic within 1e-4 of torch's on scores
    spanning +-14.3 (a strided pixel sample + the float64 sum of every map); points: index work, exact."""
    n = 0
    for gname, (h, w), aspect, moments, samples, points, stride in post_map_cases():
        scores = torch.from_numpy(load_golden(gname)["similarity_scores"]).cuda()        # (1, T, Np)
        up = model_f32.upsample_similarity(scores, (h, w), keep_aspect_ratio=aspect)[0]      # (T, H, W)
        assert tuple(up.shape) == (scores.shape[1], h, w)
        got = up.reshape(up.shape[0], -1)[:, ::stride].cpu().numpy()
        assert np.abs(got - samples).max() <= 1e-4, (gname, h, w, aspect)
        sums = up.double().sum(dim=(1, 2)).cpu().numpy()
        assert np.abs(sums - moments[:, 0]).max() <= 1e-6 * moments[:, 1].max(), (gname, h, w, aspect)
        xy = model_f32.grounding_points(scores, (h, w), keep_aspect_ratio=aspect)[0].cpu().numpy()
        assert np.array_equal(xy, points), (gname, h, w, aspect, xy, points)
        n += 1
    assert n == 10


class _Batch(dict):
    def to(self, device):
        return _Batch({k: v.to(device) for k, v in self.items()})


class _Tokenizer:
    def __call__(self, text, padding=True, truncation=True, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [[0] + [4 + (sum(map(ord, w)) % 29000) for w in t.split()] + [2] for t in texts]
        L = max(map(len, rows))
        ids = torch.tensor([r + [1] * (L - len(r)) for r in rows])
        return _Batch(input_ids=ids, attention_mask=(ids != 1).long())


class BlipImageProcessor:
    """Name-compatible stand-in (the real one needs the hub's preprocessor_config.json): grey image -> (1,3,S,S)."""

    def __init__(self, side=224):
        self.side = side

    def _prep(self, image):
        return image

    def __call__(self, image):
        image = self._prep(image.convert("L"))
        a = np.asarray(image.resize((self.side, self.side)), dtype=np.float32) / 255.0
        a = (a - 0.5) / 0.25
        return {"pixel_values": [np.stack([a, a, a])]}


class AspectRatioBlipImageProcessor(BlipImageProcessor):
    """processing.py:232-259: pad to a centred square first."""

    def _prep(self, image):
        from PIL import ImageOps
        w, h = image.size
        p = max(w, h)
        left, top = (p - w) // 2, (p - h) // 2
        return ImageOps.expand(image, border=(left, top, p - w - left, p - h - top), fill=0)


class BitImageProcessor:
    def __call__(self, image):
        raise AssertionError("never reached")


@pytest.mark.parametrize("proc_cls", [BlipImageProcessor, AspectRatioBlipImageProcessor])
def test_extract_similarity_map_follows_the_processor_branch(model_f32, oracle, tmp_path, proc_cls):
    """attention_map_base.py:12-42: returns the (H, W) map only; the crop branch follows the processor's class exactly
    as segmentation_utils.py:41/:62 decide it."""
    from PIL import Image
    from oracle.radzero_oracle import interpolate_similarity_scores
    from radzero_amd.utils import extract_similarity_map, model_inference
    rng = np.random.default_rng(3)
    path = str(tmp_path / "cxr.png")
    Image.fromarray((rng.random((300, 420)) * 255).astype(np.uint8)).save(path)
    tok, proc = _Tokenizer(), proc_cls(224)
    sim_map = extract_similarity_map(path, "There is fibrosis", model_f32, proc, tok)
    assert torch.is_tensor(sim_map) and tuple(sim_map.shape) == (300, 420)
    prob, sim_map2 = model_inference(path, "There is fibrosis", tokenizer=tok, image_processor=proc, model=model_f32)
    assert torch.equal(sim_map, sim_map2) and 0.0 < float(prob) < 1.0
    px = torch.from_numpy(np.array(proc(Image.open(path))["pixel_values"])).float()
    with torch.no_grad():
        ref = oracle.compute_logits(px, [dict(tok("There is fibrosis"))])
        ref_map = interpolate_similarity_scores(ref["similarity_scores"].reshape(-1), (300, 420),
                                                keep_aspect_ratio=proc_cls is AspectRatioBlipImageProcessor)[0]
    assert (sim_map.cpu() - ref_map).abs().max().item() <= 1e-3


def test_extract_similarity_map_rejects_other_processors(model_f32, tmp_path):
    from PIL import Image
    from radzero_amd.utils import extract_similarity_map
    path = str(tmp_path / "x.png")
    Image.fromarray(np.zeros((20, 20), np.uint8)).save(path)
    with pytest.raises(NotImplementedError):         # grounding_utils.py:248-251
        extract_similarity_map(path, "There is fibrosis", model_f32, BitImageProcessor(), _Tokenizer())


def test_from_pretrained_reads_the_reference_layout(tmp_path):
    """README.md:77-82 / inference/utils.py:24-39: a directory laid out as CxrAlignModel.save_pretrained writes it
    (config.json verbatim from the reference, tests/golden/hf_layout) -> same bits as from_state_dict."""
    from radzero_amd.checkpoint import save_checkpoint
    from radzero_amd.config import RadZeroConfig
    from radzero_amd.modeling import RadZeroModel
    from radzero_amd.weights import make_state_dict
    keys = json.load(open(os.path.join(GOLDEN_DIR, "hf_layout", "keys.json")))
    cfg = RadZeroConfig(**keys["radzero_config"])
    sd = make_state_dict(cfg, keys["weights_seed"])
    shutil.copy(os.path.join(GOLDEN_DIR, "hf_layout", "config.json"), tmp_path / "config.json")
    save_checkpoint(sd, str(tmp_path))
    px = torch.from_numpy(synthetic_pixels(2, 224, 5)).cuda()
    ids, mask = synthetic_prompts(3, 5, 9, 6)
    enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
    a = RadZeroModel.from_pretrained(str(tmp_path), torch_dtype=torch.float32, device="cuda:0")
    b = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=torch.float32, device="cuda:0")
    try:
        assert a.config == cfg
        oa, ob = a.compute_logits(px, [enc]), b.compute_logits(px, [enc])
        torch.cuda.synchronize()
        assert torch.equal(oa["logits"], ob["logits"]) and torch.equal(oa["similarity_scores"], ob["similarity_scores"])
    finally:
        a.close()
        b.close()


def test_readme_automodel_call_returns_the_hip_model(tmp_path, monkeypatch):
    """VERDICT r4 row N2 — README.md:77-82 verbatim, with a local directory in place of the hub id:
        AutoModel.from_pretrained(dir, trust_remote_code=True, torch_dtype=torch.float32, device_map=torch.device("cuda"))
    on a directory laid out by the reference's save_pretrained (tests/golden/hf_layout/config.json + synthetic weights) after
    radzero_amd.hf.export_auto_map: resolves through config.json's auto_map to the HIP model, gives the bits of from_state_dict,
    and runs the README's model_inference."""
    from PIL import Image
    from transformers import AutoModel
    from radzero_amd.checkpoint import save_checkpoint
    from radzero_amd.config import RadZeroConfig
    from radzero_amd.hf import RadZeroHFModel, export_auto_map
    from radzero_amd.modeling import RadZeroModel
    from radzero_amd.utils import model_inference
    from radzero_amd.weights import make_state_dict
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf_home"))
    monkeypatch.setenv("HF_MODULES_CACHE", str(tmp_path / "hf_home" / "modules"))
    keys = json.load(open(os.path.join(GOLDEN_DIR, "hf_layout", "keys.json")))
    cfg = RadZeroConfig(**keys["radzero_config"])
    sd = make_state_dict(cfg, keys["weights_seed"])
    ref_dir = tmp_path / "reference_layout"
    ref_dir.mkdir()
    shutil.copy(os.path.join(GOLDEN_DIR, "hf_layout", "config.json"), ref_dir / "config.json")
    save_checkpoint(sd, str(ref_dir))
    local = export_auto_map(str(ref_dir), str(tmp_path / "radzero_local"))
    assert json.load(open(ref_dir / "config.json")).get("auto_map") is None          # the original stays as the reference wrote it
    device, dtype = torch.device("cuda"), torch.float32
    model = AutoModel.from_pretrained(local, trust_remote_code=True, torch_dtype=dtype, device_map=device)
    b = RadZeroModel.from_state_dict(sd, cfg, torch_dtype=dtype, device="cuda:0")
    try:
        assert isinstance(model, RadZeroModel) and type(model).__name__ == RadZeroHFModel.__name__
        assert model.config == cfg and model.dtype == torch.float32 and model.device.type == "cuda"
        px = torch.from_numpy(synthetic_pixels(2, 224, 5)).cuda()
        ids, mask = synthetic_prompts(3, 5, 9, 6)
        enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
        oa, ob = model.compute_logits(px, [enc]), b.compute_logits(px, [enc])
        assert torch.equal(oa["logits"], ob["logits"]) and torch.equal(oa["similarity_scores"], ob["similarity_scores"])
        path = str(tmp_path / "cxr.png")
        Image.fromarray((np.random.default_rng(4).random((260, 340)) * 255).astype(np.uint8)).save(path)
        prob, sim_map = model_inference(path, "There is fibrosis", tokenizer=_Tokenizer(), image_processor=BlipImageProcessor(224), model=model)
        assert tuple(sim_map.shape) == (260, 340) and 0.0 < float(prob) < 1.0 and torch.isfinite(sim_map).all()
        # and back out: model.save_pretrained writes a directory the same AutoModel call reads (bf16 this time, device_map as a string)
        saved = model.save_pretrained(str(tmp_path / "resaved"))
        again = AutoModel.from_pretrained(saved, trust_remote_code=True, torch_dtype=torch.bfloat16, device_map="cuda")
        try:
            assert again.config == cfg and again.dtype == torch.bfloat16
            oc = again.compute_logits(px, [enc])
            assert float((oc["logits"] - oa["logits"]).abs().max()) <= 0.05
        finally:
            again.close()
    finally:
        model.close()
        b.close()


def test_text_cache_identity_level_needs_no_sync(model_f32):
    """Reference callers tokenise once and pass the same tensors for every image batch: the second call must be answered
    from the identity cache, and an in-place edit of the ids must miss it."""
    ids, mask = synthetic_prompts(4, 5, 9, 21)
    enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
    a = model_f32.encode_prompts(enc)
    n_ident = len(model_f32._text_ident_cache)
    assert model_f32.encode_prompts(enc) is a and len(model_f32._text_ident_cache) == n_ident
    enc["input_ids"][0, 1] += 1                       # in place: _version changes, content changes
    b = model_f32.encode_prompts(enc)
    assert b is not a and not torch.equal(a[0], b[0]) and torch.equal(a[1:], b[1:])
    fresh = {k: v.clone() for k, v in enc.items()}    # same content, new tensors: content level hits
    assert model_f32.encode_prompts(fresh) is b


def _rccl_child(q):
    """Runs in a FRESH process (spawn): nothing has touched the GPU before init_process_group."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
        from radzero_amd.config import RadZeroConfig
        from radzero_amd.inference import calculate_similarities
        from radzero_amd.modeling import RadZeroModel
        from radzero_amd.parallel import gather_logits, sharded_text_features
        from radzero_amd.weights import make_state_dict
        cfg = RadZeroConfig(vit_layers=2, align_layers=1, text_layers=2)
        model = RadZeroModel.from_state_dict(make_state_dict(cfg, 7), cfg, torch_dtype=torch.float32, device="cuda:0").eval()
        ids, mask = synthetic_prompts(5, 5, 9, 12)
        enc = {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}
        batches = [torch.from_numpy(synthetic_pixels(b, 224, 40 + i)) for i, b in enumerate((2, 2, 1))]
        plain_feats = model.forward_text_model(enc)["text_features_wo_l2_norm"]
        calls = []                                    # every collective issued from here on: (name, dtype, shape of the payload, device type)
        for name in ("all_gather_into_tensor", "all_gather", "gather", "all_reduce", "broadcast", "reduce_scatter_tensor", "all_to_all_single"):
            orig = getattr(dist, name)

            def spy(*a, _orig=orig, _name=name, **k):
                t = a[1] if _name == "all_gather_into_tensor" else a[0] if torch.is_tensor(a[0]) else a[1]
                calls.append((_name, str(t.dtype), list(t.shape), t.device.type))
                return _orig(*a, **k)
            setattr(dist, name, spy)
        dist_feats = sharded_text_features(lambda e: model.forward_text_model(e)["text_features_wo_l2_norm"], enc,
                                           feature_dim=cfg.hidden_size)                 # all_gather_into_tensor over RCCL
        text_calls, calls[:] = list(calls), []
        plain = calculate_similarities(batches, {"encoded_key_phrases": enc}, model)
        distd = calculate_similarities(batches, {"encoded_key_phrases": enc}, model, distributed=True)
        g = gather_logits(torch.from_numpy(plain).cuda())
        dist.barrier()
        torch.cuda.synchronize()
        q.put({"feats_equal": bool(torch.equal(plain_feats, dist_feats)), "logits_equal": bool(np.array_equal(plain, distd)),
               "gather_equal": bool(np.array_equal(g.cpu().numpy(), plain)), "shape": list(plain.shape),
               "backend": dist.get_backend(), "text_calls": text_calls, "image_loop_calls": sorted({c[0] for c in calls}),
               "image_loop_prompt_exchanges": [c[2] for c in calls if c[0] == "all_gather_into_tensor"],
               "image_loop_float_payloads": sorted({tuple(c[2]) for c in calls if "float" in c[1] and c[0] != "all_gather_into_tensor"})})
        model.close()
        dist.destroy_process_group()
    except Exception as e:      # report instead of hanging the parent
        import traceback
        q.put({"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def test_dp_driver_over_rccl_single_rank_child():
    """SURVEY.md §8(e) on hardware: a fresh child process initialises RCCL (backend "nccl", world size 1), runs
    sharded_text_features (all_gather_into_tensor), calculate_similarities(distributed=True) (size exchange + gather) and
    gather_logits, and must reproduce the non-distributed result bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_child, args=(q,))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert "error" not in res, res
    text_calls, loop_calls, payloads = res.pop("text_calls"), res.pop("image_loop_calls"), res.pop("image_loop_float_payloads")
    prompt_x = res.pop("image_loop_prompt_exchanges")
    assert res == {"feats_equal": True, "logits_equal": True, "gather_equal": True, "shape": [5, 5], "backend": "nccl"}, res
    # the prompt exchange is ONE all_gather_into_tensor of (ceil(T / W), 768) fp32 on the device — 5 x 768 x 4 B here
    assert text_calls == [["all_gather_into_tensor", "torch.float32", [5, 768], "cuda"]] or text_calls == [("all_gather_into_tensor", "torch.float32", [5, 768], "cuda")], text_calls
    # ... and the batch driver (calculate_similarities(distributed=True), 3 image batches) issues that prompt exchange exactly ONCE (cached
    # for the later batches) and otherwise NO data-path collective: only the result gather of (rows, T) logits, row counts exchanged first
    assert prompt_x == [[5, 768]], prompt_x
    assert set(loop_calls) <= {"all_gather", "gather", "all_gather_into_tensor"}, loop_calls
    assert all(len(pl) == 2 and pl[1] == 5 for pl in payloads), payloads          # float payloads are (rows, T) logits, never tokens / maps
    assert p.exitcode == 0
