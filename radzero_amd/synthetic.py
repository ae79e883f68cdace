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
