"""World-size-2 (and 3) gloo tests of the data-parallel plumbing on CPU: prompt sharding + ONE all_gather of
text embeddings rebuilds the exact table every rank would compute alone; logits gather keeps image order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from radzero_amd.parallel import gather_logits, shard_range, sharded_text_features


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _encode(enc):
    """Deterministic stand-in for the text encoder: row-wise function of (ids, mask) only."""
    ids = enc["input_ids"].double()
    m = enc["attention_mask"].double()
    base = (ids * m).sum(1, keepdim=True) / m.sum(1, keepdim=True).clamp(min=1)
    return torch.cat([torch.sin(base * (k + 1) * 0.013) for k in range(8)], dim=1).float()


def _worker(rank, world, port, n_prompts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        ids = torch.randint(4, 30000, (n_prompts, 9), generator=g)
        mask = (torch.rand(n_prompts, 9, generator=g) > 0.2).long()
        mask[:, 0] = 1
        enc = {"input_ids": ids, "attention_mask": mask}
        table = sharded_text_features(_encode, enc)
        ref = _encode(enc)
        ok_table = torch.equal(table, ref) and torch.equal(sharded_text_features(_encode, enc, feature_dim=8), ref)
        local = torch.full((2, n_prompts), float(rank)) + torch.arange(n_prompts).float() * 0.01
        allg = gather_logits(local)
        ok_gather = True
        if rank == 0:
            ok_gather = allg.shape == (2 * world, n_prompts) and all(
                torch.equal(allg[2 * r:2 * r + 2], torch.full((2, n_prompts), float(r)) + torch.arange(n_prompts).float() * 0.01)
                for r in range(world))
        else:
            ok_gather = allg is None
        q.put((rank, ok_table, ok_gather))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_prompts", [(2, 14), (2, 1), (3, 14), (3, 2), (3, 4), (4, 9)])
def test_sharded_text_features_gloo(world, n_prompts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_prompts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok_t and ok_g for _, ok_t, ok_g in res), res


class _FakeModel:
    """CPU stand-in with the model protocol calculate_similarities uses (device, config.hidden_size,
    forward_text_model / encode_prompts / compute_logits): logits are a deterministic function of (image, prompt)."""

    class config:
        hidden_size = 8

    device = torch.device("cpu")

    def forward_text_model(self, enc):
        return {"text_features_wo_l2_norm": _encode(enc)}

    def encode_prompts(self, enc):
        return _encode(enc)

    def compute_logits(self, pixel_values, encoded_key_phrases, text_features=None, **_):
        img = pixel_values.double().mean(dim=(1, 2, 3))                       # (B,)
        return {"logits": (img[:, None] * text_features.double().sum(1)[None, :] + text_features.double()[:, 0][None, :]).float()}


def _batches(n_images, batch):
    g = torch.Generator().manual_seed(5)
    px = torch.randn(n_images, 3, 4, 4, generator=g)
    return [px[i:i + batch] for i in range(0, n_images, batch)]


class _LoggingDataset:
    """Map-style dataset of (3, 4, 4) float images that records which items were read (the stand-in for decoding a file)."""

    def __init__(self, n_images):
        g = torch.Generator().manual_seed(5)
        self.px = torch.randn(n_images, 3, 4, 4, generator=g)
        self.read = []

    def __len__(self):
        return self.px.shape[0]

    def __getitem__(self, i):
        self.read.append(int(i))
        return self.px[i]


def _driver_worker(rank, world, port, n_images, batch, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radzero_amd.inference import calculate_similarities
        from radzero_amd.parallel import StridedBatchSampler
        g = torch.Generator().manual_seed(1)
        enc = {"input_ids": torch.randint(4, 30000, (5, 7), generator=g), "attention_mask": torch.ones(5, 7, dtype=torch.long)}
        ds = _LoggingDataset(n_images)
        if mode == "dataset":
            got = calculate_similarities(ds, {"encoded_key_phrases": enc}, _FakeModel(), distributed=True, batch_size=batch)
        elif mode == "iterable":           # this rank's own batches, built with the same sampler (what a torch DataLoader(batch_sampler=...) would yield)
            mine = (torch.stack([ds[j] for j in idxs]) for idxs in StridedBatchSampler(n_images, batch, rank, world))
            got = calculate_similarities(mine, {"encoded_key_phrases": enc}, _FakeModel(), distributed=True, presharded=True)
        elif mode == "full_iterable":      # the contract of rounds 1-3: every rank passes the same full sequence, the driver deals it
            full = [torch.stack([ds.px[j] for j in range(i, min(i + batch, n_images))]) for i in range(0, n_images, batch)]
            got = calculate_similarities(full, {"encoded_key_phrases": enc}, _FakeModel(), distributed=True)
            ds.read = None                  # nothing is read through the dataset in this mode
        q.put((rank, None if got is None else got.tolist(), None if ds.read is None else sorted(ds.read)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["dataset", "iterable", "full_iterable"])
@pytest.mark.parametrize("world,n_images,batch", [(2, 7, 2), (3, 7, 2), (3, 2, 2), (2, 5, 8), (4, 9, 1)])
def test_batch_driver_shards_the_source_and_restores_order(world, n_images, batch, mode):
    """the SOURCE is sharded, not the results — every rank reads exactly the items of its own batches (batch i
    belongs to rank i % world; uneven shares, some ranks empty), no item is read twice, and rank 0 gets exactly the single-process
    result in the original image order."""
    from radzero_amd.inference import calculate_similarities
    g = torch.Generator().manual_seed(1)
    enc = {"input_ids": torch.randint(4, 30000, (5, 7), generator=g), "attention_mask": torch.ones(5, 7, dtype=torch.long)}
    want = calculate_similarities(_batches(n_images, batch), {"encoded_key_phrases": enc}, _FakeModel())
    assert want.shape == (n_images, 5)
    assert np.array_equal(calculate_similarities(_LoggingDataset(n_images), {"encoded_key_phrases": enc}, _FakeModel(), batch_size=batch), want)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, n_images, batch, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (got, read) for r, got, read in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r][0] is None for r in range(1, world))
    assert np.array_equal(np.asarray(res[0][0], np.float32), want)
    for r in range(world):
        share = [j for i in range(r, (n_images + batch - 1) // batch, world) for j in range(i * batch, min((i + 1) * batch, n_images))]
        assert res[r][1] is None or res[r][1] == share, (r, res[r][1], share)          # its own items, each exactly once, nothing else


def test_strided_batch_sampler_partitions_the_batches():
    from radzero_amd.parallel import StridedBatchSampler
    for n, bs, w in ((7, 2, 2), (64, 8, 8), (5, 8, 2), (0, 4, 3), (9, 1, 4)):
        per_rank = [list(StridedBatchSampler(n, bs, r, w)) for r in range(w)]
        assert [len(StridedBatchSampler(n, bs, r, w)) for r in range(w)] == [len(p) for p in per_rank]
        nb = (n + bs - 1) // bs
        merged = [per_rank[i % w][i // w] for i in range(nb)]
        assert [j for b in merged for j in b] == list(range(n))
    with pytest.raises(ValueError):
        StridedBatchSampler(4, 0)


def test_shard_range_covers_everything():
    for n in (1, 2, 14, 64, 193):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n)), (n, w, spans)


def test_single_process_passthrough():
    enc = {"input_ids": torch.randint(4, 100, (5, 6)), "attention_mask": torch.ones(5, 6, dtype=torch.long)}
    assert torch.equal(sharded_text_features(_encode, enc), _encode(enc))
    x = torch.randn(3, 4)
    assert gather_logits(x) is x


def _ckpt_worker(rank, world, port, q):
    """bench.py's one-checkpoint-per-node path: local rank 0 generates + writes, the others read after a barrier."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from radzero_amd.config import RadZeroConfig
        from radzero_amd.weights import make_state_dict, state_dict_digest
        cfg = RadZeroConfig(vit_layers=1, align_layers=1, text_layers=1, vocab_size=300)
        sd = bench.node_shared_state_dict(cfg, 77, rank, True)
        ref = make_state_dict(cfg, 77)
        same = set(sd) == set(ref) and all(np.array_equal(np.asarray(sd[k], np.float32).reshape(-1), np.asarray(ref[k], np.float32).reshape(-1)) for k in ref)
        q.put((rank, bool(same), state_dict_digest({k: np.asarray(v, np.float32).reshape(np.shape(ref[k])) for k, v in sd.items()}) == state_dict_digest(ref)))
    finally:
        dist.destroy_process_group()


def test_bench_checkpoint_is_built_once_per_node():
    """N ranks of one node must not each generate the synthetic checkpoint; every rank ends up with the same tensors
    as the generator gives, and the temporary file is gone afterwards (gloo, world size 3)."""
    import glob
    import tempfile
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True, True) for r in range(world)], res
    assert not glob.glob(os.path.join(tempfile.gettempdir(), f"radzero_bench_ckpt_77_{port}*"))


class _SlowDataset(_LoggingDataset):
    """`__getitem__` takes `delay` seconds: a file decode."""

    def __init__(self, n_images, delay):
        super().__init__(n_images)
        self.delay = delay

    def __getitem__(self, i):
        import time
        time.sleep(self.delay)
        return super().__getitem__(i)


class _SlowModel(_FakeModel):
    """compute_logits takes `delay` seconds: a forward the host waits for."""

    def __init__(self, delay):
        self.delay = delay

    def compute_logits(self, pixel_values, encoded_key_phrases, text_features=None, **kw):
        import time
        time.sleep(self.delay)
        return super().compute_logits(pixel_values, encoded_key_phrases, text_features=text_features, **kw)


def test_dataset_reads_overlap_the_forward():
    """with a map-style dataset source the items of batches k + 1, k + 2 are read by a background thread
    while batch k computes — a slow `__getitem__` does not serialise with the step — and the result, the read order and the error
    behaviour are those of the plain loop."""
    import time
    from radzero_amd.inference import calculate_similarities
    g = torch.Generator().manual_seed(1)
    enc = {"input_ids": torch.randint(4, 30000, (5, 7), generator=g), "attention_mask": torch.ones(5, 7, dtype=torch.long)}
    n, bs, item_s, fwd_s = 24, 4, 0.02, 0.08                      # 6 batches: 0.08 s of reads and 0.08 s of forward each
    want = calculate_similarities(_batches(n, bs), {"encoded_key_phrases": enc}, _FakeModel())
    t0 = time.perf_counter()
    ds = _SlowDataset(n, item_s)
    got = calculate_similarities(ds, {"encoded_key_phrases": enc}, _SlowModel(fwd_s), batch_size=bs)
    t_overlap = time.perf_counter() - t0
    t0 = time.perf_counter()
    ds2 = _SlowDataset(n, item_s)
    got2 = calculate_similarities(ds2, {"encoded_key_phrases": enc}, _SlowModel(fwd_s), batch_size=bs, overlap=False)
    t_serial = time.perf_counter() - t0
    assert np.array_equal(got, want) and np.array_equal(got2, want)
    assert ds.read == list(range(n)) and ds2.read == list(range(n))
    serial_floor = n * item_s + (n // bs) * fwd_s                 # 0.96 s: reads + forwards back to back
    assert t_serial >= 0.95 * serial_floor
    assert t_overlap <= 0.75 * t_serial, (t_overlap, t_serial)   # ideal: one batch of reads + 6 forwards = 0.56 s

    class _Broken(_LoggingDataset):
        def __getitem__(self, i):
            if i == 9:
                raise OSError("unreadable file")
            return super().__getitem__(i)

    with pytest.raises(OSError, match="unreadable file"):          # the producer's exception reaches the caller
        calculate_similarities(_Broken(n), {"encoded_key_phrases": enc}, _FakeModel(), batch_size=bs)


class _FakeAlignmentModel(_FakeModel):
    """The protocol of a global_alignment model (modeling.py:330-353): a text projector doubles the text feature width."""

    class config:
        hidden_size = 4
        use_text_projection = True

    def forward_text_model(self, enc):
        return {"text_features_wo_l2_norm": _encode(enc)}            # (t, 8) = 2 x hidden

    def compute_logits(self, pixel_values, encoded_key_phrases, text_features=None, **_):
        assert text_features is not None and text_features.shape[1] == 8      # the driver hands the gathered table through
        return super().compute_logits(pixel_values, encoded_key_phrases, text_features=text_features)


def _alignment_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radzero_amd.inference import calculate_similarities
        g = torch.Generator().manual_seed(1)
        enc = {"input_ids": torch.randint(4, 30000, (2, 7), generator=g), "attention_mask": torch.ones(2, 7, dtype=torch.long)}
        got = calculate_similarities(_LoggingDataset(7), {"encoded_key_phrases": enc}, _FakeAlignmentModel(), distributed=True, batch_size=2)
        q.put((rank, None if got is None else got.tolist()))
    finally:
        dist.destroy_process_group()


def test_batch_driver_with_projected_text_features_gloo():
    """with use_text_projection the text features are 2 x hidden wide; with fewer prompts (2) than ranks (3) one rank pads
    its empty shard by `feature_dim` — which has to be the projected width, or the all_gather buffers disagree."""
    from radzero_amd.inference import calculate_similarities
    g = torch.Generator().manual_seed(1)
    enc = {"input_ids": torch.randint(4, 30000, (2, 7), generator=g), "attention_mask": torch.ones(2, 7, dtype=torch.long)}
    want = calculate_similarities(_batches(7, 2), {"encoded_key_phrases": enc}, _FakeAlignmentModel())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_alignment_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None and res[2] is None and np.array_equal(np.asarray(res[0], np.float32), want)


# ---- batch shaping (round 6): host logic only -------------------------------------------------------------------------------------------
def _pad_rule(nv, batch=0):
    """csrc/api.hip rz_model::pad_tokens, default rule (test-side restatement: the product asks the library, rz_padded_tokens)"""
    p128, p256 = (nv + 127) // 128 * 128, (nv + 255) // 256 * 256
    if (p256 - nv) * 50 <= nv:
        return p256
    odd = p128 != p256 and (p256 - p128) * 10 <= p128
    return p256 if odd and (batch == 0 or ((batch & 1) and batch * p256 >= 4 * 256)) else p128


def test_preferred_batch_picks_whole_tile_rounds():
    from radzero_amd.shaping import gemm_tile_cost, preferred_batch
    # the reference's evaluation point: 518^2 (N = 1370), batch 64 -> 62 (341 row tiles x 3 = 1023 tiles = 4 rounds of 256; 64: 1056 -> 5)
    assert preferred_batch(64, 1370, _pad_rule) == 62
    assert gemm_tile_cost(62, 1370, _pad_rule) < 0.92 * gemm_tile_cost(64, 1370, _pad_rule)
    # the headline shape is already whole rounds (672 row tiles: 7.9 / 23.6 / 31.5 rounds): untouched; so are tiny batches and shapes below the model's range
    assert preferred_batch(32, 5330, _pad_rule) == 32 and preferred_batch(16, 5330, _pad_rule) == 16
    assert preferred_batch(1, 11882, _pad_rule) == 1 and preferred_batch(4, 1370, _pad_rule) == 4 and preferred_batch(64, 257, _pad_rule) == 64
    for b in (8, 24, 40, 64, 96, 128, 200):
        for n in (257, 1370, 5330):
            p = preferred_batch(b, n, _pad_rule)
            assert b - max(2, b // 16) <= p <= b                       # never larger than the caller's batch, never far below it


class _ShapedModel(_FakeModel):
    def __init__(self, target):
        super().__init__()
        self.target, self.forward_sizes = target, []

    def preferred_batch(self, batch, height, width):
        return self.target if batch > self.target else batch

    def compute_logits(self, pixel_values, encoded_key_phrases, text_features=None, **kw):
        self.forward_sizes.append(int(pixel_values.shape[0]))
        return super().compute_logits(pixel_values, encoded_key_phrases, text_features=text_features, **kw)


def test_reshape_batches_keeps_order_rows_and_results():
    """`_reshape_batches` re-cuts 64, 64, 64, 10 images into forwards of 62, 62, 62, 16: the first 62 images of a source batch go out as a VIEW of it (no
    copy), the leftovers gather into a later forward, and `ranges` says where every forward's rows belong — `_place_rows` restores the source order.
    The SOURCE batch sizes are what the distributed interleave receives; a size that is already good passes through untouched."""
    import numpy as np
    import torch

    from radzero_amd.inference import _place_rows, _reshape_batches
    sizes = [64, 64, 64, 10]
    tag = 0
    batches = []
    for s in sizes:
        batches.append(torch.arange(tag, tag + s, dtype=torch.float32).view(s, 1, 1, 1).expand(s, 3, 2, 2).contiguous())
        tag += s
    m = _ShapedModel(62)
    rows = []
    out = list(_reshape_batches(iter(batches), m, rows, enabled=True))
    assert [int(o.shape[0]) for o, _ in out] == [62, 62, 62, 16] and rows == sizes
    for k in range(3):                                               # steady state: views of the source batches, not copies
        assert out[k][0].data_ptr() == batches[k].data_ptr() and out[k][1] == [(64 * k, 62)]
    assert out[3][1] == [(62, 2), (126, 2), (190, 2), (192, 10)]
    placed = _place_rows([(o[:, 0, 0, :1] * 1.0, r) for o, r in out], 1, torch.device("cpu"))       # "logits" = the image's tag
    assert np.array_equal(placed[:, 0].numpy(), np.arange(202, dtype=np.float32))
    # many batches: the carry reaches the target and goes out as a forward of its own (one copy per 31 source batches at 64 -> 62)
    many = [batches[0]] * 33
    got = list(_reshape_batches(iter(many), _ShapedModel(62), [], enabled=True))
    assert [int(o.shape[0]) for o, _ in got] == [62] * 31 + [62] + [62, 62] + [4]
    assert sum(c for _, r in got for _, c in r) == 33 * 64 and sorted(f for _, r in got for f, _ in r)[0] == 0
    rows2 = []
    same = list(_reshape_batches(iter(batches), _ShapedModel(64), rows2, enabled=True))
    assert all(a is b for (a, _), b in zip(same, batches)) and rows2 == sizes and [r for _, r in same] == [[(0, 64)], [(64, 64)], [(128, 64)], [(192, 10)]]
    rows3 = []
    off = list(_reshape_batches(iter(batches), m, rows3, enabled=False))
    assert all(a is b for (a, _), b in zip(off, batches)) and rows3 == sizes
    # a change of resolution inside the stream sends what was carried out first
    mixed = [batches[0], torch.zeros((5, 3, 4, 4)), batches[1]]
    got = list(_reshape_batches(iter(mixed), _ShapedModel(62), [], enabled=True))
    assert [tuple(g.shape) for g, _ in got] == [(62, 3, 2, 2), (2, 3, 2, 2), (5, 3, 4, 4), (62, 3, 2, 2), (2, 3, 2, 2)]
    assert [r for _, r in got] == [[(0, 62)], [(62, 2)], [(64, 5)], [(69, 62)], [(131, 2)]]


def _shaped_driver_worker(rank, world, port, n_images, batch, target, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radzero_amd.inference import calculate_similarities
        g = torch.Generator().manual_seed(1)
        enc = {"input_ids": torch.randint(4, 30000, (5, 7), generator=g), "attention_mask": torch.ones(5, 7, dtype=torch.long)}
        m = _ShapedModel(target)
        got = calculate_similarities(_LoggingDataset(n_images), {"encoded_key_phrases": enc}, m, distributed=True, batch_size=batch)
        q.put((rank, None if got is None else got.tolist(), m.forward_sizes))
    finally:
        dist.destroy_process_group()


def test_batch_shaping_under_the_distributed_driver():
    """Batch shaping + source sharding together (what an N > 1 GPU run of calculate_similarities does): every rank re-cuts ITS OWN stream of batches
    (batch 8 -> forwards of 6 + the carried leftovers), rank 0 still receives exactly the single-process, un-shaped result in source order."""
    from radzero_amd.inference import calculate_similarities
    world, n_images, batch, target = 2, 45, 8, 6
    g = torch.Generator().manual_seed(1)
    enc = {"input_ids": torch.randint(4, 30000, (5, 7), generator=g), "attention_mask": torch.ones(5, 7, dtype=torch.long)}
    want = calculate_similarities(_batches(n_images, batch), {"encoded_key_phrases": enc}, _FakeModel())
    shaped_single = calculate_similarities(_batches(n_images, batch), {"encoded_key_phrases": enc}, _ShapedModel(target))
    assert np.array_equal(shaped_single, want)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shaped_driver_worker, args=(r, world, port, n_images, batch, target, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(np.array(res[0][1], np.float32), want) and res[1][1] is None
    # rank 0 owns batches 0, 2, 4 (8 + 8 + 8 images) -> 6, 6, 6 as views and the 6 carried; rank 1 batches 1, 3, 5 (8 + 8 + 5) -> 6, 6, then 9 left
    assert res[0][2] == [6, 6, 6, 6] and res[1][2] == [6, 6, 6, 3]
