"""Batch shaping: which forward size a batch DRIVER should use (pure host arithmetic, no GPU).

The large-shape GEMM (csrc/gemm8.hip) is persistent: 256 workgroups, one per CU, walk the 256 x 256 output tiles of a launch, so a launch
costs WHOLE rounds of 256 tiles.  At the reference's own evaluation point (518^2, batch 64: exp/cxr_pt/configs/radzero.yaml:19,
exp/cxr_pt/config.yaml:55) the two N = 768 GEMMs of a block have 352 x 3 = 1056 tiles = 4.125 rounds -> 5; 62 images have 341 x 3 = 1023 =
3.996 -> 4, and 62 images per forward measured 2478 against 2408 images per second (profiles/NOTEBOOK.md r5).  The reference's drivers take
the DataLoader's batch as it comes (exp/cxr_pt/inference/utils.py:81-100); `radzero_amd.inference.calculate_similarities` re-cuts the
stream of batches into forwards of `preferred_batch(...)` images — results and order unchanged (images are independent on this path)."""
from __future__ import annotations

from typing import Callable, Optional

PERSISTENT_WORKGROUPS = 256      # gemm8.hip: one 160 KB workgroup per CU
MIN_ROWS = 128 * 256             # below 128 row tiles the 128 x 128 kernel may take some launches (gemm.hip big_tiles_pay): the model does not price those


def gemm_tile_cost(batch: int, n_tokens: int, padded_tokens: Callable[[int, int], int], hidden: int = 768, mlp_ratio: int = 4) -> float:
    """Relative GEMM time PER IMAGE of one forward of `batch` images: the four linear shapes of a block (q|k|v, out-proj, fc1, fc2) as
    256 x 256 tiles in whole rounds of 256, a round costing ~K.  inf where the large-tile kernel does not apply."""
    if batch <= 0:
        return float("inf")
    rows = batch * padded_tokens(n_tokens, batch)
    if rows % 256 or rows < MIN_ROWS:
        return float("inf")
    d, f, g = hidden, hidden * mlp_ratio, PERSISTENT_WORKGROUPS
    cost = 0
    for n_out, k in ((3 * d, d), (d, d), (f, d), (d, f)):
        tiles = (rows // 256) * (n_out // 256)
        cost += -(-tiles // g) * g * k
    return cost / batch


def preferred_batch(batch: int, n_tokens: int, padded_tokens: Callable[[int, int], int], hidden: int = 768, mlp_ratio: int = 4,
                    max_drop: Optional[int] = None, min_gain: float = 0.02) -> int:
    """The size in [batch - max_drop, batch] (default max_drop = max(2, batch / 16)) with the lowest modelled GEMM cost per image, if the whole
    step is predicted >= `min_gain` faster than at `batch` (the GEMMs' share of the step from the FLOP counts: linear 24 N D^2 against
    attention 4 N^2 D per block); else `batch`.  Never larger than `batch`: the caller's workspace was sized for it."""
    if batch < 8 or n_tokens <= 1:
        return batch
    base = gemm_tile_cost(batch, n_tokens, padded_tokens, hidden, mlp_ratio)
    if base == float("inf"):
        return batch
    drop = max(2, batch // 16) if max_drop is None else max_drop
    best, best_cost = batch, base
    for b in range(batch - 1, max(batch - drop, 1) - 1, -1):
        c = gemm_tile_cost(b, n_tokens, padded_tokens, hidden, mlp_ratio)
        if c < best_cost:
            best, best_cost = b, c
    gemm_share = 24.0 * hidden / (24.0 * hidden + 4.0 * n_tokens)
    return best if gemm_share * (1.0 - best_cost / base) >= min_gain else batch
