"""Batch drivers mirroring exp/cxr_pt/inference/utils.py:42-106 (`process_class_prompts`,
`calculate_similarities`) and grounding_utils.py:31-66 (`get_similarity_scores`) — SURVEY.md §8(f) rank 2.

Differences from the reference, all deliberate:
  * prompt embeddings are encoded ONCE per prompt set and cached (the reference re-encodes every prompt for
    every image batch, modeling.py:290-307) — same numbers, T x fewer text forwards per batch;
  * the unused negative-prompt tokenisation (utils.py:57-62) is kept only for signature compatibility;
  * under torch.distributed the SOURCE is sharded — batches are dealt round-robin to the ranks by a strided batch
    sampler, a rank never reads an image it does not compute — and logits are gathered to rank 0 in the original order;
  * raw detector images are preprocessed on the device (the raw bytes cross PCIe, not fp32 pixels), overlapped with the
    previous batch's forward; a map-style dataset is read by a background thread two batches ahead (the reference's
    4-worker DataLoader, inference/utils.py:81-90, does the same job with processes).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .parallel import StridedBatchSampler, gather_row_shards, interleave_row_shards, sharded_text_features


def process_class_prompts(text_prompt: Dict[str, List[str]], tokenizer, model):
    prompts = [text_prompt[str(i)][0] for i in range(len(text_prompt))]
    negatives = [p.replace("There is", "There is no") for p in prompts]
    enc = tokenizer(prompts, padding=True, truncation=True, return_tensors="pt").to(model.device)
    neg = tokenizer(negatives, padding=True, truncation=True, return_tensors="pt").to(model.device)
    return {"encoded_key_phrases": enc, "encoded_negative_phrases": neg}


def _prefetch(gen, depth: int = 2):
    """Run the generator `gen` in a background thread, `depth` results ahead (bounded queue): host-side dataset reads / decodes of batch
    k + 1, k + 2 overlap the compute of batch k.  Exceptions of the producer are re-raised at the consumer; if the consumer stops
    early the producer is told to stop and unblocked."""
    import queue
    import threading
    q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
    stop = threading.Event()
    END = object()

    def put(item) -> bool:
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def run():
        try:
            for item in gen:
                if not put((item, None)):
                    return
            put((END, None))
        except BaseException as e:          # noqa: BLE001 — handed to the consumer
            put((END, e))

    t = threading.Thread(target=run, name="radzero-prefetch", daemon=True)
    t.start()
    try:
        while True:
            item, err = q.get()
            if item is END:
                if err is not None:
                    raise err
                return
            yield item
    finally:
        stop.set()


def _is_dataset(source) -> bool:
    return hasattr(source, "__getitem__") and hasattr(source, "__len__") and not torch.is_tensor(source) and not isinstance(source, (list, tuple))


def _collate_default(items, model, preprocessor):
    """A batch of dataset items -> pixel_values on the model's device.  Raw detector images (2-D / HWC integer or float arrays, sizes may
    differ) go through the DEVICE preprocessing (radzero_amd.preprocess.DevicePreprocessor: cv2 min-max, Pillow-exact bicubic, rescale,
    normalise) — 7.2 MB per 2048 x 1760 uint16 image over PCIe instead of 12.6 MB of fp32 pixels at 1024^2; items that already are
    (3, S, S) float tensors are stacked and copied."""
    import numpy as np
    first = items[0]
    if torch.is_tensor(first) and first.dim() == 3 and first.shape[0] == 3 and first.is_floating_point():
        return torch.stack([t.float() for t in items]).to(model.device, non_blocking=True)
    if preprocessor is None:
        raise ValueError("calculate_similarities: raw images need preprocessor=DevicePreprocessor(size, ...) (or pass collate_fn=)")
    raws = []
    for it in items:
        a = it if torch.is_tensor(it) else torch.from_numpy(np.ascontiguousarray(np.asarray(it)))
        raws.append(a.pin_memory() if (a.device.type == "cpu" and model.device.type == "cuda") else a)
    return preprocessor(raws)


def _reshape_batches(batches, model, src_rows: list, enabled: bool):
    """Batch shaping: re-cut a stream of pixel batches into forwards of `model.preferred_batch(...)` images where the caller's
    batch size leaves the persistent GEMM's last round of 256 tiles nearly empty (518^2 x 64 -> forwards of 62: +3 % images per second).
    Yields (pixel_values, ranges): `ranges` = [(first row in the stream, count), ...] says where the forward's rows belong in the source order.
    Copy-free in the steady state: the first `target` images of every source batch go out as a VIEW of that batch; the few images left over are
    carried, and once `target` of them have gathered they go out as one forward (one copy of `target` images per ~target / leftover source
    batches instead of one per batch); the tail goes out as one last, smaller forward.  `src_rows` receives the SOURCE batch sizes (what the
    distributed interleave needs).  Images are independent on this path: no result bit changes in the 16-bit modes; in the fp32 mode the operand
    form follows the forward's size as it always has.  Runs in the caller's stream context (the side stream of
    `calculate_similarities` when it overlaps)."""
    carry, carried, target, pos = [], 0, None, 0          # carry: (tensor, first row) pieces in stream order

    def drain(limit):
        """pieces of the carry adding up to `limit` rows -> (tensor, ranges)"""
        nonlocal carry, carried
        take, ranges, got = [], [], 0
        while got < limit:
            t, first = carry[0]
            k = min(int(t.shape[0]), limit - got)
            take.append(t[:k])
            ranges.append((first, k))
            got += k
            if k == int(t.shape[0]):
                carry.pop(0)
            else:
                carry[0] = (t[k:], first + k)
        carried -= limit
        return (torch.cat(take, dim=0) if len(take) > 1 else take[0]), ranges

    for px in batches:
        n = int(px.shape[0])
        src_rows.append(n)
        if enabled and px.dim() == 4 and target is None:
            pick = getattr(model, "preferred_batch", None)
            target = int(pick(n, int(px.shape[2]), int(px.shape[3]))) if pick is not None else n
            if target >= n:
                enabled = False                  # the caller's size is already a good one: everything passes through untouched
        if not enabled or px.dim() != 4:
            yield px, [(pos, n)]
            pos += n
            continue
        if carry and (carry[0][0].shape[1:] != px.shape[1:] or carry[0][0].dtype != px.dtype):
            yield drain(carried)                 # the resolution changed: what was carried goes out first
        px = px.to(model.device, non_blocking=True)
        off = 0
        while n - off >= target:
            yield px[off:off + target], [(pos + off, target)]
            off += target
        if off < n:
            carry.append((px[off:], pos + off))
            carried += n - off
        pos += n
        while carried >= target:
            yield drain(target)
    if carried:
        yield drain(carried)


def _place_rows(pieces, n_prompts, device):
    """[(logits (k, T), ranges)] -> (rows, T) in source order"""
    total = sum(c for _, ranges in pieces for _, c in ranges)
    if not pieces:
        return torch.zeros((0, n_prompts), dtype=torch.float32, device=device)
    in_order = all(len(r) == 1 for _, r in pieces) and all(pieces[i][1][0][0] + pieces[i][1][0][1] == pieces[i + 1][1][0][0] for i in range(len(pieces) - 1))
    if in_order:
        return torch.cat([lg for lg, _ in pieces], dim=0)
    out = torch.empty((total, pieces[0][0].shape[1]), dtype=pieces[0][0].dtype, device=pieces[0][0].device)
    for lg, ranges in pieces:
        off = 0
        for first, count in ranges:
            out[first:first + count] = lg[off:off + count]
            off += count
    return out


_guard_warned = False


def _warn_guard_reruns(model, n_batches: int) -> None:
    """fp32 mode: a checkpoint whose activations leave the f16 planes' range makes every forward run twice, the second time on the
    exact-fp32 kernels at about a quarter of the speed — results are right, throughput is not.  Said once per process."""
    global _guard_warned
    reruns = getattr(model, "guard_reruns", None)
    if _guard_warned or reruns is None or getattr(model, "dtype", None) is not torch.float32:
        return
    n = int(reruns())
    if n > 0:
        import warnings
        _guard_warned = True
        warnings.warn(f"radzero_amd fp32 mode: {n} forward(s) so far (this call: {n_batches} batch(es)) left the range of the f16 operand planes and were "
                      "repeated on the exact-fp32 kernels (option f32_split_guard; correct results at ~1/4 of the speed). "
                      "The default MX form carries activations up to |x| = 1792 (e4m3 hi plane); model.set_model_option('gemm_f32_mx', 0) selects the three-plane "
                      "f16 form (|x| up to 65504, ~20 % slower than MX, 4 x faster than exact); model.set_model_option('gemm_f32_split', 0) + ('attn_f32_split', 0) "
                      "runs the exact kernels directly.", RuntimeWarning, stacklevel=3)


@torch.no_grad()
def calculate_similarities(source, text_batch, model, distributed: bool = False, *, batch_size: Optional[int] = None,
                           collate_fn: Optional[Callable] = None, preprocessor=None, overlap: bool = True, presharded: bool = False,
                           batch_shaping: bool = True):
    """Class logits (n_images, T) as float32 numpy (exp/cxr_pt/inference/utils.py:70-106, :103-104), in the order of the source.

    `source` is either
      * a map-style dataset (`__len__` + `__getitem__`, e.g. the reference's InferDataset, inference/dataset.py:14-28) together with
        `batch_size`: the driver forms the batches itself — items [i * batch_size, (i + 1) * batch_size) — and, under
        torch.distributed, rank r takes batches r, r + world, ... through a StridedBatchSampler, so that a rank NEVER reads or decodes
        an image it will not compute (the host-side share of an 8-GPU run is 1/8 per rank, not 8/8).  A batch of items becomes
        pixel_values through `collate_fn(items)` if given, else: (3, S, S) float tensors are stacked; raw images (2-D / HWC uint8 /
        uint16 / float arrays, sizes may differ) run the device preprocessing `preprocessor` (DevicePreprocessor) — the raw bytes
        cross PCIe, not fp32 pixels.  With `overlap`, a background thread reads the items two batches ahead (a slow `__getitem__` —
        file decode — runs beside the forward, not between launches) and, with a CUDA model, batch k + 1 is copied and preprocessed
        on a side stream while batch k computes (`overlap=False`: everything on the calling thread and stream — for a dataset whose
        `__getitem__` must not run on another thread);
      * or an iterable of pixel_values tensors (B, 3, S, S).  Under torch.distributed every rank passes the SAME full sequence of
        batches and computes batch i where i % world == rank (the others are skipped: the contract of rounds 1-3).  `presharded=True`
        says the iterable holds THIS RANK'S batches only (global batch i = the (i // world)-th batch of rank i % world — e.g. a
        DataLoader built on StridedBatchSampler): nothing is skipped, a rank never produces a batch it does not compute.  Passing a
        full sequence with presharded=True would compute everything on every rank and return world x duplicated rows — hence opt-in.

    batch_shaping (default on): where the incoming batch size leaves the persistent GEMM's last tile round nearly empty, the stream of batches is
    re-cut into forwards of model.preferred_batch(...) images (never larger than the caller's batch; `_reshape_batches`) — same results, same
    order, a few per cent more images per second at e.g. 518^2 x 64 (the reference's own evaluation batch, exp/cxr_pt/config.yaml:55).

    distributed=True (one process per GPU, torch.distributed initialised): the prompt set is encoded once, sharded over ranks + one
    all_gather (parallel.sharded_text_features); per-rank logits — unequal row counts are fine, a rank may get nothing — are gathered to
    rank 0 and interleaved back into the source order there.  Ranks other than 0 return None."""
    enc = text_batch["encoded_key_phrases"]
    dist_on = distributed and dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist_on else (1, 0)
    encode = lambda e: model.forward_text_model(e)["text_features_wo_l2_norm"]
    # with a text projector (cls_alignment / global_alignment heads, modeling.py:70-73) the text features are 2 x hidden wide
    feat_dim = model.config.hidden_size * (2 if getattr(model.config, "use_text_projection", False) else 1)
    feats = sharded_text_features(encode, enc, feature_dim=feat_dim) if dist_on else model.encode_prompts(enc)
    n_prompts = int(feats.shape[0])
    cuda = model.device.type == "cuda"

    if _is_dataset(source):
        if not batch_size or batch_size <= 0:
            raise ValueError("calculate_similarities: a dataset source needs batch_size")
        make = collate_fn if collate_fn is not None else (lambda items: _collate_default(items, model, preprocessor))
        fetched = ([source[j] for j in idxs] for idxs in StridedBatchSampler(len(source), batch_size, rank, world))
        if overlap:
            fetched = _prefetch(fetched, depth=2)           # host reads / decodes beside the compute
        batches = (make(items) for items in fetched)        # device work (H2D, preprocessing) stays on the calling thread's side stream
    elif presharded or world == 1:
        batches = iter(source)
    else:
        batches = (px for i, px in enumerate(source) if i % world == rank)

    out, rows = [], []          # out: (logits, where its rows belong); rows: SOURCE batch sizes (the distributed interleave deals source batches)
    batches = _reshape_batches(batches, model, rows, enabled=bool(batch_shaping))      # a model without preferred_batch() passes through untouched
    if cuda and overlap:
        # produce batch k + 1 (dataset reads, H2D, device preprocessing) on a side stream while batch k computes on the current one
        main = torch.cuda.current_stream(model.device)
        side = torch.cuda.Stream(device=model.device)

        def produce():
            with torch.cuda.stream(side):
                try:
                    px, ranges = next(batches)
                except StopIteration:
                    return None
                px = px.to(model.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            return px, ranges, ev

        nxt = produce()
        while nxt is not None:
            px, ranges, ev = nxt
            main.wait_event(ev)
            px.record_stream(main)
            logits = model.compute_logits(pixel_values=px, encoded_key_phrases=[enc], text_features=feats)["logits"]
            out.append((logits.reshape(px.shape[0], -1).clone(), ranges))
            nxt = produce()
    else:
        for px, ranges in batches:
            logits = model.compute_logits(pixel_values=px.to(model.device), encoded_key_phrases=[enc], text_features=feats)["logits"]
            out.append((logits.reshape(px.shape[0], -1), ranges))
    logits = _place_rows(out, n_prompts, feats.device)
    _warn_guard_reruns(model, len(rows))
    if dist_on:
        all_rows = [None] * world
        dist.all_gather_object(all_rows, rows)
        shards = gather_row_shards(logits.float())
        if shards is None:
            return None
        ordered = interleave_row_shards(shards, all_rows)
        logits = torch.cat(ordered, dim=0) if ordered else shards[0]
    return logits.float().cpu().numpy()


@torch.no_grad()
def get_similarity_scores(batches: Iterable[torch.Tensor], text_batch, model):
    """grounding_utils.py:31-66: keeps similarity_scores (B, T, Np) on the device, concatenated over batches."""
    enc = text_batch["encoded_key_phrases"]
    feats = model.encode_prompts(enc)
    return torch.cat([model.compute_logits(pv.to(model.device), [enc], text_features=feats)["similarity_scores"].clone()
                      for pv in batches], dim=0)
