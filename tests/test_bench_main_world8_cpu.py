"""bench.main() ITSELF at world size 8: the first real 8-GPU run is the driver's, and every world > 1 branch of main()
— node_shared_state_dict(multi=True), sharded_text_features with EMPTY ranks (14 prompts over 8 ranks: ceil = 2 per rank, ranks 7 has none),
the closing barrier, all_reduce(MAX) of the elapsed time, all_gather_into_tensor of the per-rank rates, rank-0-only emission, the `rccl`
evidence block — has so far only run at world 1 on a GPU or piecewise under gloo.  Here: 8 processes over gloo on CPU, device and backend injected
through RZ_BENCH_DEVICE / RZ_BENCH_BACKEND, RadZeroModel replaced by a tiny CPU stub with the same method surface (the product model has no CPU
path and is not what is tested here: the host logic of bench.main() is)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8


class StubModel:
    """The slice of RadZeroModel's surface bench.main() touches."""
    text_cache_enabled = True

    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        self._prof = False
        self._launches = 0
        self.text_calls = []

    @classmethod
    def from_state_dict(cls, sd, cfg, torch_dtype=None, device=None):
        assert "a" in sd and sd["a"].shape == (10,), "every rank must receive the node-shared checkpoint"
        return cls(cfg, device)

    def eval(self):
        return self

    def set_f32_precision(self, level):
        pass

    def forward_text_model(self, enc):
        import torch
        t = enc["input_ids"].shape[0]
        self.text_calls.append(t)
        # deterministic in the token ids, so that the gathered table can be checked against an un-sharded encode
        return {"text_features_wo_l2_norm": enc["input_ids"].float().sum(1, keepdim=True).repeat(1, self.cfg.hidden_size) * 1e-3 if t else torch.zeros((0, self.cfg.hidden_size))}

    def compute_logits(self, px, encs, text_features=None):
        import torch
        assert text_features is not None and text_features.shape == (14, self.cfg.hidden_size)
        if self._prof:
            self._launches += self.cfg.num_blocks
        b = px.shape[0]
        return {"logits": torch.zeros((b, 14)) + text_features[:, 0][None], "similarity_scores": torch.zeros((b, 14, 4))}

    def profile(self, enable, families=None):
        self._prof = bool(enable)
        if enable:
            self._launches = 0

    def profile_read(self):
        fams = ("attn", "gemm", "rowops", "vlcabs", "post")
        return {f: {"ms": 1.0 * self._launches if f in ("attn", "gemm") else 0.1, "launches": self._launches} for f in fams}

    def get_model_option(self, name):
        return 0

    def guard_reruns(self):
        return 0

    def close(self):
        pass


def _worker(rank, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RZ_BENCH_DEVICE="cpu", RZ_BENCH_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import io
    from contextlib import redirect_stdout

    import bench
    import radzero_amd.modeling as modeling
    import radzero_amd.parallel as parallel
    modeling.RadZeroModel = StubModel                                  # bench.main() imports the name from the module at call time
    bench.make_state_dict = lambda cfg, seed: {"a": np.arange(10, dtype=np.float32)}      # the 840 MB synthetic checkpoint is not what this test is about
    seen = []
    real = parallel.sharded_text_features

    def spy(fn, enc, **kw):
        table = real(fn, enc, **kw)
        seen.append(table.clone())
        return table
    parallel.sharded_text_features = spy
    sys.argv = ["bench.py", "--gpus", str(WORLD), "--steps", "3", "--warmup", "1", "--batch", "2", "--side", "224"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"stdout": buf.getvalue(), "table_col0": seen[0][:, 0].tolist(), "table_shape": list(seen[0].shape)}, f)


@pytest.mark.timeout(600)
def test_bench_main_at_world_8_over_gloo(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from bench import free_port
    from radzero_amd.synthetic import synthetic_prompts
    port = free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(WORLD)]
    # rank-0-only emission: ONE JSON line from rank 0, nothing from the others
    lines = [ln for ln in recs[0]["stdout"].splitlines() if ln.strip()]
    assert len(lines) == 1, recs[0]["stdout"]
    assert all(r["stdout"].strip() == "" for r in recs[1:])
    res = json.loads(lines[0])
    assert res["n_gpus"] == WORLD and res["scaling"] == "weak" and res["steps"] == 3 and res["warmup"] == 1
    assert res["config"]["global_batch"] == WORLD * 2 and res["config"]["parallelism"] == "dp8"
    assert res["value"] == pytest.approx(WORLD * 2 * 3 / (res["ms_per_step"] * 3e-3), rel=1e-3)     # whole-job rate from the MAX-over-ranks time
    rc = res["rccl"]
    assert rc["backend"] == "gloo" and rc["world_size"] == WORLD
    assert rc["data_path_collectives"] == 0 and rc["prompt_exchange_collectives"] == 1 and rc["timed_region_barriers"] == 1
    assert rc["all_gather_bytes"] == WORLD * 2 * 768 * 4                                            # ceil(14 / 8) = 2 rows per rank, fp32
    assert len(rc["per_rank_images_per_s"]["all"]) == WORLD and all(v > 0 for v in rc["per_rank_images_per_s"]["all"])
    assert res["value"] <= sum(rc["per_rank_images_per_s"]["all"]) * 1.001                          # the slowest rank bounds the job
    assert "cpu_baseline" not in res and "other_configs" not in res                                  # N = 1 only
    assert res["roofline"]["launches"] == 3 * 14                                                     # events of the timed steps only
    # every rank holds the same, complete prompt table — including the ranks that encoded nothing (14 prompts over 8 ranks)
    ids, _ = synthetic_prompts(14, 6, 10, 4321)
    want = (ids.astype(np.float32).sum(1) * 1e-3).tolist()
    for r in recs:
        assert r["table_shape"] == [14, 768]
        assert r["table_col0"] == pytest.approx(want, rel=1e-6)
