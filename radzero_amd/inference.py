"""Batch drivers mirroring exp/cxr_pt/inference/utils.py:42-106 (`process_class_prompts`,
`calculate_similarities`) and grounding_utils.py:31-66 (`get_similarity_scores`) — SURVEY.md §8(f) rank 2.

Differences from the reference, all deliberate:
  * prompt embeddings are encoded ONCE per prompt set and cached (the reference re-encodes every prompt for
    every image batch, modeling.py:290-307) — same numbers, T x fewer text forwards per batch;
  * the unused negative-prompt tokenisation (utils.py:57-62) is kept only for signature compatibility;
  * under torch.distributed the batches are dealt round-robin to the ranks and logits are gathered to rank 0
    in the original order.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .parallel import gather_row_shards, sharded_text_features


def process_class_prompts(text_prompt: Dict[str, List[str]], tokenizer, model):
    prompts = [text_prompt[str(i)][0] for i in range(len(text_prompt))]
    negatives = [p.replace("There is", "There is no") for p in prompts]
    enc = tokenizer(prompts, padding=True, truncation=True, return_tensors="pt").to(model.device)
    neg = tokenizer(negatives, padding=True, truncation=True, return_tensors="pt").to(model.device)
    return {"encoded_key_phrases": enc, "encoded_negative_phrases": neg}


@torch.no_grad()
def calculate_similarities(batches: Iterable[torch.Tensor], text_batch, model, distributed: bool = False):
    """`batches` yields pixel_values tensors (B, 3, S, S) (the reference builds them with its DataLoader +
    collate_fn, inference/dataset.py:31-51).  Returns class logits (n_images, T) as float32 numpy
    (utils.py:103-104) in the order of `batches`.

    distributed=True (one process per GPU, torch.distributed initialised): EVERY rank passes the same `batches`
    sequence; rank r computes the batches with index i % world == r (round-robin, so a streaming loader needs no
    length), the prompt set is encoded once, sharded over ranks + one all_gather (parallel.sharded_text_features), and
    the per-rank logits — unequal row counts are fine, a rank may get nothing — are gathered to rank 0 and put back
    into `batches` order there.  Ranks other than 0 return None."""
    enc = text_batch["encoded_key_phrases"]
    dist_on = distributed and dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist_on else (1, 0)
    encode = lambda e: model.forward_text_model(e)["text_features_wo_l2_norm"]
    feats = sharded_text_features(encode, enc, feature_dim=model.config.hidden_size) if dist_on else model.encode_prompts(enc)
    n_prompts = int(feats.shape[0])
    out, rows_per_batch = [], []
    for i, pixel_values in enumerate(batches):
        rows_per_batch.append(int(pixel_values.shape[0]))
        if i % world != rank:
            continue
        out.append(model.compute_logits(pixel_values=pixel_values.to(model.device), encoded_key_phrases=[enc],
                                        text_features=feats)["logits"].reshape(pixel_values.shape[0], -1))
    logits = torch.cat(out, dim=0) if out else torch.zeros((0, n_prompts), dtype=torch.float32, device=feats.device)
    if dist_on:
        shards = gather_row_shards(logits.float())
        if shards is None:
            return None
        # undo the round-robin: batch i is the (i // world)-th batch of rank i % world
        offsets = [0] * world
        ordered = []
        for i, b in enumerate(rows_per_batch):
            r = i % world
            ordered.append(shards[r][offsets[r]: offsets[r] + b])
            offsets[r] += b
        logits = torch.cat(ordered, dim=0) if ordered else shards[0]
    return logits.float().cpu().numpy()


@torch.no_grad()
def get_similarity_scores(batches: Iterable[torch.Tensor], text_batch, model):
    """grounding_utils.py:31-66: keeps similarity_scores (B, T, Np) on the device, concatenated over batches."""
    enc = text_batch["encoded_key_phrases"]
    feats = model.encode_prompts(enc)
    return torch.cat([model.compute_logits(pv.to(model.device), [enc], text_features=feats)["similarity_scores"].clone()
                      for pv in batches], dim=0)
