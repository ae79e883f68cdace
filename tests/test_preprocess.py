"""Image preprocessing: host tables vs Pillow itself (CPU), device byte pipeline vs Pillow + numpy (GPU).
Byte/integer work: bit-exact against PIL.Image.resize(BICUBIC) of the 8-bit image."""
import numpy as np
import pytest
import torch
from PIL import Image

from radzero_amd.preprocess import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, resample_tables


def _resample_numpy(img8, out_w, out_h):
    """Apply the tables exactly as Pillow's 8-bit passes do (horizontal first, uint8 intermediate)."""
    h, w = img8.shape[:2]
    x = img8.reshape(h, w, -1).astype(np.int64)
    bh, kh, _ = resample_tables(w, out_w)
    tmp = np.zeros((h, out_w, x.shape[2]), np.int64)
    for xx in range(out_w):
        lo, n = bh[xx]
        tmp[:, xx] = np.clip(((1 << 21) + (x[:, lo:lo + n] * kh[xx, :n, None]).sum(1)) >> 22, 0, 255)
    bv, kv, _ = resample_tables(h, out_h)
    out = np.zeros((out_h, out_w, x.shape[2]), np.int64)
    for yy in range(out_h):
        lo, n = bv[yy]
        out[yy] = np.clip(((1 << 21) + (tmp[lo:lo + n] * kv[yy, :n, None, None]).sum(0)) >> 22, 0, 255)
    return out.astype(np.uint8).reshape((out_h, out_w) + img8.shape[2:])


@pytest.mark.parametrize("shape,out", [((300, 420), 224), ((97, 61), 128), ((64, 64), 224), ((1200, 1000), 518)])
def test_tables_reproduce_pillow_bicubic(shape, out):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((out, out), resample=Image.BICUBIC))
    got = _resample_numpy(img, out, out)
    assert np.array_equal(got, ref)


def test_tables_rgb_matches_pillow():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(150, 200, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((96, 96), resample=Image.BICUBIC))
    assert np.array_equal(_resample_numpy(img, 96, 96), ref)


def _reference_pipeline(raw, size, minmax=True):
    """dataset.py:31-51 + Blip processor semantics with numpy/PIL (cv2 is absent: NORM_MINMAX restated)."""
    a = raw.astype(np.float64)
    if minmax:
        lo, hi = a.min(), a.max()
        scale = 255.0 / (hi - lo) if hi - lo > 2.220446049250313e-16 else 0.0
        a8 = np.clip(np.rint(a * scale - lo * scale), 0, 255).astype(np.uint8)
    else:
        a8 = raw.astype(np.uint8)
    pil = Image.fromarray(a8).convert("RGB").resize((size, size), resample=Image.BICUBIC)
    x = np.asarray(pil).astype(np.float32) * np.float32(1.0 / 255.0)
    x = (x - np.array(OPENAI_CLIP_MEAN, np.float32)) / np.array(OPENAI_CLIP_STD, np.float32)
    return np.transpose(x, (2, 0, 1))[None]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,shape,size", [(np.uint16, (512, 400), 224), (np.uint8, (300, 420), 224), (np.float32, (257, 129), 266),
                                               (np.uint8, (64, 80, 3), 224), (np.uint16, (2048, 1760), 1024)])
def test_device_preprocess_matches_reference_pipeline(dtype, shape, size):
    from radzero_amd.preprocess import DevicePreprocessor
    rng = np.random.default_rng(shape[0] + size)
    if dtype == np.float32:
        raw = (rng.standard_normal(shape) * 300 + 1000).astype(np.float32)
    else:
        raw = rng.integers(0, 4096 if dtype == np.uint16 else 256, size=shape).astype(dtype)
    if raw.ndim == 3:       # RGB uint8 goes through without min-max (already 8 bit)... and with it
        pre = DevicePreprocessor(size, minmax_normalize=True)
    else:
        pre = DevicePreprocessor(size, minmax_normalize=True)
    t = torch.from_numpy(raw.astype(np.int32) if dtype == np.uint16 else raw)
    out = pre(t).cpu().numpy()
    assert np.array_equal(out, pre.single_image_entry(t).cpu().numpy())          # batched kernels == the one-image entry point, bit for bit
    ref = _reference_pipeline(raw, size, True)
    assert out.shape == ref.shape == (1, 3, size, size)
    assert np.abs(out - ref).max() <= 2e-6          # identical bytes; float rescale/normalise rounding only


@pytest.mark.gpu
def test_device_preprocess_constant_image_and_errors():
    from radzero_amd.preprocess import DevicePreprocessor
    pre = DevicePreprocessor(224)
    out = pre(torch.full((100, 100), 7, dtype=torch.uint8)).cpu().numpy()
    ref = _reference_pipeline(np.full((100, 100), 7, np.uint8), 224)
    assert np.abs(out - ref).max() <= 2e-6            # constant image -> all zeros after min-max (scale 0)
    with pytest.raises(ValueError):
        pre(torch.zeros(10, 10, 2, dtype=torch.uint8))


# ---- the REAL image processors (tests/golden/preprocess_blip.npz, tools/make_goldens_preprocess.py): transformers' BlipImageProcessor
# and the reference's own AspectRatioBlipImageProcessor, run in the build container on 8-bit images (cv2's min-max step itself is not
# installed anywhere: unpinned, restated in radzero_amd.synthetic.minmax_to_u8) --------------------------------------------------------
import os
import zlib

from conftest import GOLDEN_DIR
from radzero_amd.synthetic import minmax_to_u8, synthetic_cxr_raw


def _pre_cases():
    z = np.load(os.path.join(GOLDEN_DIR, "preprocess_blip.npz"), allow_pickle=False)
    stride = int(z["sample_stride"])
    for name in [str(n) for n in z["cases"]]:
        code, h, w, seed, size, aspect = (int(v) for v in z[name + "|meta"])
        yield name, ("uint8", "uint16")[code], (h, w), seed, size, bool(aspect), z[name + "|moments"], z[name + "|samples"], int(z[name + "|u8_crc"][0]), z[name + "|mean_std"], stride


def _pad_square(a8):
    h, w = a8.shape
    side = max(h, w)
    out = np.zeros((side, side), np.uint8)
    out[(side - h) // 2:(side - h) // 2 + h, (side - w) // 2:(side - w) // 2 + w] = a8
    return out


@pytest.mark.parametrize("case", list(_pre_cases()), ids=lambda c: c[0])
def test_host_tables_and_padding_reproduce_the_real_processors(case):
    """CPU: min-max -> [pad_to_square] -> the host's Pillow tables applied in numpy give EXACTLY the bytes the real processor resized to
    (CRC32 of the uint8 image recovered from its fp32 output): pins resample_tables, the pass order, and the padding arithmetic."""
    name, dtype, shape, seed, size, aspect, _, _, crc, _, _ = case
    a8 = minmax_to_u8(synthetic_cxr_raw(dtype, shape, seed))
    if aspect:
        a8 = _pad_square(a8)
    got = _resample_numpy(a8, size, size)
    rgb = np.repeat(got[None], 3, axis=0)                       # grey -> RGB replication, channel-first like pixel_values
    assert zlib.crc32(rgb.tobytes()) == crc, name


@pytest.mark.gpu
@pytest.mark.parametrize("aspect", [False, True])
def test_device_batch_matches_the_real_processors(aspect):
    """GPU: ONE batched call per processor kind over all of its fixture images (different sizes and dtypes in the same batch, 8-bit and
    16-bit; for the AspectRatio processor: tall, wide and odd-difference paddings) against the real processors' outputs: identical
    bytes after the resize, float rescale / normalise within 2e-6, float64 moments."""
    from radzero_amd.preprocess import DevicePreprocessor
    cases = [c for c in _pre_cases() if c[5] == aspect]
    for size in sorted({c[4] for c in cases}):
        group = [c for c in cases if c[4] == size]
        pre = DevicePreprocessor(size, keep_aspect_ratio=aspect)
        raws = [synthetic_cxr_raw(c[1], c[2], c[3]) for c in group]
        out = pre([torch.from_numpy(r.astype(np.int32) if r.dtype == np.uint16 else r) for r in raws]).cpu().numpy()
        assert out.shape == (len(group), 3, size, size)
        for o, c in zip(out, group):
            name, _, _, _, _, _, moments, samples, crc, mean_std, stride = c
            assert np.abs(o.reshape(3, -1)[:, ::stride] - samples).max() <= 2e-6, name
            m = np.stack([o.astype(np.float64).sum((1, 2)), (o.astype(np.float64) ** 2).sum((1, 2))], 1)
            assert np.abs(m - moments).max() <= 2e-6 * size * size * 4, name
            u8 = np.rint((o.astype(np.float64) * mean_std[1][:, None, None] + mean_std[0][:, None, None]) * 255.0).astype(np.uint8)
            assert zlib.crc32(u8.tobytes()) == crc, name


@pytest.mark.gpu
def test_device_batch_feeds_the_model_and_rejects_bad_batches(cfg, state_dict):
    """The batch's output tensor is directly rz_vision_forward's input; ragged batches keep per-image independence (an image alone ==
    the same image inside a batch, bit for bit); bad descriptors are refused by the C-ABI, not by a fault."""
    import ctypes
    from radzero_amd import _lib
    from radzero_amd.config import RadZeroConfig
    from radzero_amd.modeling import RadZeroModel
    from radzero_amd.preprocess import DevicePreprocessor
    from radzero_amd.weights import make_state_dict
    pre = DevicePreprocessor(224)
    raws = [synthetic_cxr_raw("uint16", (300 + 17 * i, 280 - 9 * i), 100 + i) for i in range(5)]
    ts = [torch.from_numpy(r.astype(np.int32)) for r in raws]
    batch = pre(ts)
    for i in (0, 3):
        assert torch.equal(pre(ts[i])[0], batch[i])
    small = RadZeroConfig(vit_layers=1, align_layers=1, text_layers=1)
    m = RadZeroModel.from_state_dict(make_state_dict(small, 3), small, torch_dtype=torch.float32, device="cuda:0").eval()
    try:
        toks = m.forward_vision_model(batch)["vision_tokens"]
        assert toks.shape == (5, 257, 768) and torch.isfinite(toks).all()
    finally:
        m.close()
    with pytest.raises(ValueError):
        pre([])
    with pytest.raises(ValueError):
        pre([torch.zeros(10, 10, 2, dtype=torch.uint8)])
    lib = _lib.load()
    d = (_lib.RzImageDesc * 1)()
    assert lib.rz_preprocess_batch_workspace(d, 1, 224) == 0                      # null image / zero shape
    mean = (ctypes.c_float * 3)(0, 0, 0)
    assert lib.rz_preprocess_batch(d, 1, 224, mean, mean, 1.0, 1, ctypes.c_void_p(batch.data_ptr()), 16, ctypes.c_void_p(batch.data_ptr()), None) == 10001


@pytest.mark.gpu
def test_batch_driver_from_raw_images_on_a_side_stream(cfg, state_dict):
    """The data-parallel driver's default input path: a map-style dataset of RAW uint16 detector images of different
    sizes -> calculate_similarities(dataset, batch_size=, preprocessor=): 37 items in batches of 16 (> 16 descriptors per launch group, a
    short last batch), dataset reads + H2D + device preprocessing on a side stream while the previous batch computes.  Must equal, bit for
    bit, preprocessing and computing the same batches one after the other on the current stream, and the un-overlapped driver."""
    from radzero_amd.config import RadZeroConfig
    from radzero_amd.inference import calculate_similarities
    from radzero_amd.modeling import RadZeroModel
    from radzero_amd.preprocess import DevicePreprocessor
    from radzero_amd.synthetic import synthetic_prompts
    from radzero_amd.weights import make_state_dict

    class Raws:
        def __init__(self):
            self.items = [synthetic_cxr_raw("uint16", (260 + 7 * (i % 5), 300 - 11 * (i % 3)), 300 + i) for i in range(37)]
            self.read = []

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            self.read.append(i)
            return self.items[i]

    small = RadZeroConfig(vit_layers=1, align_layers=1, text_layers=1)
    m = RadZeroModel.from_state_dict(make_state_dict(small, 3), small, torch_dtype=torch.float32, device="cuda:0").eval()
    try:
        ids, mask = synthetic_prompts(4, 5, 9, 12)
        tb = {"encoded_key_phrases": {"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda()}}
        pre = DevicePreprocessor(224)
        ds = Raws()
        got = calculate_similarities(ds, tb, m, batch_size=16, preprocessor=pre)
        assert got.shape == (37, 4) and ds.read == list(range(37))
        plain = calculate_similarities(Raws(), tb, m, batch_size=16, preprocessor=pre, overlap=False)
        want = []
        for lo in range(0, 37, 16):
            px = pre([torch.from_numpy(r) for r in ds.items[lo:lo + 16]])
            want.append(m.compute_logits(px, [tb["encoded_key_phrases"]])["logits"].reshape(px.shape[0], -1).cpu().numpy())
        want = np.concatenate(want)
        assert np.array_equal(got, want) and np.array_equal(plain, want)
        with pytest.raises(ValueError):
            calculate_similarities(Raws(), tb, m, batch_size=16)          # raw items need a preprocessor (or a collate_fn)
    finally:
        m.close()
