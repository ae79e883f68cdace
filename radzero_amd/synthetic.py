"""Deterministic synthetic inputs: post-normalisation CXR pixels ~ N(0,1) and
tokenised prompts with MPNet's special ids (BOS 0, EOS 2, PAD 1).  numpy Philox => identical bits
in the build container and on the GPU box, so goldens only need to store seeds."""
from __future__ import annotations

import numpy as np

BOS, PAD, EOS = 0, 1, 2


def synthetic_pixels(batch: int, side: int, seed: int = 1234, channels: int = 3) -> np.ndarray:
    rng = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0x5052]))
    return rng.standard_normal(size=(batch, channels, side, side), dtype=np.float32)


def synthetic_prompts(n_prompts: int, min_len: int = 6, max_len: int = 10, seed: int = 4321,
                      vocab_size: int = 30527):
    """Returns (input_ids, attention_mask) int64 (T, L) with L = longest prompt (tokenizer padding=True).
    Lengths include BOS/EOS; content ids ~ U[4, 30000)."""
    rng = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0x7478]))
    lens = rng.integers(min_len, max_len + 1, size=n_prompts)
    lens[rng.integers(0, n_prompts)] = max_len
    L = int(lens.max())
    ids = np.full((n_prompts, L), PAD, np.int64)
    mask = np.zeros((n_prompts, L), np.int64)
    for i, n in enumerate(lens):
        n = int(n)
        ids[i, 0] = BOS
        ids[i, 1:n - 1] = rng.integers(4, min(30000, vocab_size), size=n - 2)
        ids[i, n - 1] = EOS
        mask[i, :n] = 1
    return ids, mask


def synthetic_cxr_raw(dtype: str, shape, seed: int) -> np.ndarray:
    """A raw chest-X-ray-like image as a detector / DICOM delivers it (before any preprocessing): smooth anatomy-like field + noise,
    8-bit or 12-bit-in-uint16.  Inputs of the preprocessing fixture (tools/make_goldens_preprocess.py) and of `bench.py --raw-images`."""
    rng = np.random.default_rng(seed)
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    field = 0.5 + 0.3 * np.sin(xx / w * 5.1 + 0.3) * np.cos(yy / h * 3.7) + 0.15 * np.exp(-(((xx - 0.6 * w) / (0.2 * w)) ** 2 + ((yy - 0.4 * h) / (0.3 * h)) ** 2))
    field += rng.standard_normal((h, w)) * 0.04
    top = 255 if dtype == "uint8" else 4095
    return np.clip(field * top, 3 if dtype == "uint8" else 40, top - (5 if dtype == "uint8" else 77)).astype(dtype)


def minmax_to_u8(raw: np.ndarray) -> np.ndarray:
    """cv2.normalize(raw, None, 0, 255, NORM_MINMAX, CV_8U) restated (cv2 is not installed anywhere this runs: UNPINNED):
    saturate(round_half_even(v * 255 / (max - min) - min * 255 / (max - min))), all channels together, scale 0 for a constant image."""
    a = raw.astype(np.float64)
    lo, hi = a.min(), a.max()
    scale = 255.0 / (hi - lo) if hi - lo > 2.220446049250313e-16 else 0.0
    return np.clip(np.rint(a * scale - lo * scale), 0, 255).astype(np.uint8)
