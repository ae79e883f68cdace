"""Data-parallel inference over the GPUs of one node: one process per GPU (torch.distributed; backend
"nccl" IS RCCL on ROCm, "gloo" in the CPU tests).

The path shards over images with no data-path collective: each rank runs the vision
encoder + VL-CABS on its own images.  The only exchange is the one-time prompt encoding: prompts are
sharded T/W per rank, encoded, and ONE all_gather of the (T/W, 768) fp32 embeddings rebuilds the (T, 768)
table on every rank (21.5 KB at T=14: latency-bound on xGMI, done once per prompt set and cached).
The reference has no counterpart (its inference is rank-0 only, exp/cxr_pt/run.py:135).
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n_items for `rank`; items are padded to a multiple of world."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


class StridedBatchSampler:
    """Index lists of the batches THIS rank computes: the global sequence of batches (items [i * batch_size, (i + 1) * batch_size) of a
    map-style dataset, in order, last one short) is dealt round-robin — rank r takes batches r, r + world, r + 2 world, ... — so a rank
    never touches (reads, decodes, collates) an item it will not compute.  Usable as `batch_sampler=` of a torch DataLoader (the
    reference builds its loader in inference/utils.py:81-90) or directly (inference.calculate_similarities does)."""

    def __init__(self, n_items: int, batch_size: int, rank: int = 0, world: int = 1):
        if batch_size <= 0 or world <= 0 or not 0 <= rank < world:
            raise ValueError("StridedBatchSampler: batch_size > 0 and 0 <= rank < world")
        self.n_items, self.batch_size, self.rank, self.world = int(n_items), int(batch_size), int(rank), int(world)
        self.n_batches = (self.n_items + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(self.rank, self.n_batches, self.world):
            yield list(range(i * self.batch_size, min((i + 1) * self.batch_size, self.n_items)))

    def __len__(self):
        return len(range(self.rank, self.n_batches, self.world))


def interleave_row_shards(shards, rows_per_batch):
    """Undo the round-robin deal: shards[r] = rank r's rows (its batches concatenated in its own order), rows_per_batch[r] = the row
    count of each of its batches; global batch i is the (i // world)-th batch of rank i % world."""
    world = len(shards)
    offsets = [0] * world
    ordered = []
    for k in range(max((len(c) for c in rows_per_batch), default=0)):
        for r in range(world):
            if k < len(rows_per_batch[r]):
                b = rows_per_batch[r][k]
                ordered.append(shards[r][offsets[r]: offsets[r] + b])
                offsets[r] += b
    return ordered


def sharded_text_features(encode_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                          encoded: Dict[str, torch.Tensor], group=None, feature_dim: int | None = None) -> torch.Tensor:
    """Encode this rank's share of the prompts with `encode_fn` ((t,L) ids/mask -> (t, D) fp32) and
    all_gather the rest.  Every rank returns the full (T, D) table in prompt order.
    `feature_dim` (= hidden size) lets ranks that receive no prompt (T not a multiple of the world size) size their
    padding without an extra collective; when omitted it is agreed on with one small all_reduce."""
    if not (dist.is_available() and dist.is_initialized()):
        return encode_fn(encoded)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ids, mask = encoded["input_ids"], encoded["attention_mask"]
    t = ids.shape[0]
    per = (t + world - 1) // world
    lo, hi = shard_range(t, rank, world)
    dev = ids.device if ids.is_cuda else torch.device("cpu")
    local = encode_fn({"input_ids": ids[lo:hi], "attention_mask": mask[lo:hi]}).float() if hi > lo else None
    d = feature_dim
    if d is None:
        some_rank_empty = per * (world - 1) >= t          # same on every rank: the collective below is entered by all or none
        if some_rank_empty:
            dim = torch.tensor([0 if local is None else local.shape[1]], dtype=torch.int64, device=dev)
            dist.all_reduce(dim, op=dist.ReduceOp.MAX, group=group)
            d = int(dim.item())
        else:
            d = local.shape[1]
    buf = torch.zeros((per, d), dtype=torch.float32, device=dev)
    if local is not None:
        buf[: hi - lo] = local.to(dev)
    out = torch.empty((world * per, d), dtype=torch.float32, device=dev)
    if dev.type == "cuda":
        dist.all_gather_into_tensor(out, buf, group=group)
    else:
        _all_gather_cpu(out, buf, group)
    return out[:t].contiguous()


def _all_gather_cpu(out, buf, group):
    parts = [torch.empty_like(buf) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, buf, group=group)
    out.copy_(torch.cat(parts, dim=0))


def gather_logits(local_logits: torch.Tensor, group=None, dst: int = 0):
    """Optional result gather of the (B_local, T) logits to rank `dst` (similarity maps stay sharded).  Ranks may hold
    different numbers of rows (uneven image shards, ranks with none): row counts are exchanged first, rows are padded
    to the largest shard for the gather and trimmed again on `dst`.  Returns the rank-ordered concatenation on `dst`,
    None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_logits
    parts = gather_row_shards(local_logits, group=group, dst=dst)
    return torch.cat(parts, dim=0) if parts is not None else None


def gather_row_shards(local: torch.Tensor, group=None, dst: int = 0):
    """List of every rank's (rows_r, T) tensor on rank `dst` (None elsewhere); rows_r may differ per rank and be 0."""
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local.device
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if nmax == 0:
        return [local[:0] for _ in range(world)] if rank == dst else None
    padded = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, parts, dst=dst, group=group)
    return [p[:c] for p, c in zip(parts, counts)] if parts is not None else None
