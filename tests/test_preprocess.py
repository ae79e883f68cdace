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
