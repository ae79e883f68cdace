"""`python bench.py --gpus N` outside torch.distributed.run starts its own ranks: the launcher branch is driven here
with a stub child (no GPU, no torch.cuda call in the parent)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = [sys.executable, os.path.join(ROOT, "tests", "bench_stub_child.py")]


def _run(extra, timeout=120, env_extra=None):
    env = dict(os.environ, RZ_BENCH_CHILD_CMD=json.dumps(STUB))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True, timeout=timeout, env=env)


def test_launcher_starts_n_ranks_and_relays_one_json_line():
    r = _run(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                     # ONE JSON line on stdout, everything else on stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 4 and rec["argv"] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert rec["ipc"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") and rec["launched_by"] == "bench.py"
    seen = sorted(l for l in r.stderr.splitlines() if "stub rank" in l)
    assert [l.split("stub rank ")[1].split(" ")[0] for l in seen] == ["0/4", "1/4", "2/4", "3/4"], r.stderr
    assert len({l.split(" port ")[1].split(" ")[0] for l in seen}) == 1            # one rendezvous port for all ranks
    assert all(" addr 127.0.0.1 " in l for l in seen)
    assert all(f"local {i} " in seen[i] for i in range(4))


def test_launcher_reports_a_failed_rank_and_does_not_retry():
    t0 = time.time()
    r = _run(["--gpus", "2"], env_extra={"RZ_STUB_FAIL": "1"})
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert r.stdout.strip() == ""                        # no bench line from a run with a failed rank
    assert "rank 1 exited with code 3" in r.stderr and r.stderr.count("stub rank 1/2") == 1
    assert time.time() - t0 < 60


def test_launcher_terminates_ranks_left_waiting_for_a_dead_one():
    src = f"import bench, sys; sys.exit(bench.launch_ranks(2, [], child_cmd={STUB!r}, grace_s=1.0))"
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env=dict(os.environ, RZ_STUB_FAIL="0", RZ_STUB_HANG="1"))
    assert r.returncode != 0 and "terminating rank 1" in r.stderr, r.stderr
    assert time.time() - t0 < 60


def test_launcher_overall_timeout_when_every_rank_hangs():
    """ranks that all deadlock without any of them exiting must not hang the launcher forever."""
    src = f"import bench, sys; sys.exit(bench.launch_ranks(2, [], child_cmd={STUB!r}, grace_s=1.0, overall_timeout_s=3.0))"
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env=dict(os.environ, RZ_STUB_HANG="all"))
    assert r.returncode != 0 and "no rank has finished after 3 s" in r.stderr and "terminating rank 0" in r.stderr and "terminating rank 1" in r.stderr, r.stderr
    assert r.stdout.strip() == "" and time.time() - t0 < 60


def test_visible_gpu_count_loads_no_gpu_library():
    """the launcher parent counts GPUs from the KFD topology (or skips the check), never through torch.cuda / HIP."""
    src = "import bench, torch; n = bench.visible_gpu_count(); assert n is None or n >= 0; assert not torch.cuda.is_initialized(); print('ok', n)"
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr


def test_single_rank_through_the_launcher():
    r = _run(["--gpus", "1", "--launch", "--force-dist"])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def _clog_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    from radzero_amd.parallel import sharded_text_features
    clog = bench.CollectiveLog()                      # before init_process_group, as bench.main() does
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ids = torch.arange(14 * 6).reshape(14, 6)
        enc = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
        clog.phase = "prompt_exchange"
        table = sharded_text_features(lambda e: e["input_ids"].float().repeat(1, 128), enc, feature_dim=768)      # (14, 768) fp32
        clog.phase = "setup"
        dist.barrier()
        clog.phase = "timed_steps"
        for _ in range(3):
            (table * 2).sum()                          # a "step": no collective
        clog.phase = "closing_barrier"
        dist.barrier()
        clog.phase = "after"
        t = torch.tensor([1.0])
        dist.all_reduce(t)
        q.put((rank, clog.count("prompt_exchange"), clog.bytes("prompt_exchange", "all_gather"), clog.count("timed_steps"),
               sum(1 for ph, n, _ in clog.calls if ph == "closing_barrier" and n == "barrier"), tuple(table.shape), clog.count("after")))
    finally:
        dist.destroy_process_group()


def test_collective_log_counts_what_the_process_group_carried():
    """The `rccl` block of the bench line is READ from the live process group: the collective log around
    torch.distributed sees ONE gather for the prompt exchange — world x ceil(T / world) x 768 fp32 — nothing inside the timed steps, and the
    contract's closing barrier.  gloo, world size 2, CPU."""
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_clog_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n_exchange, gather_bytes, n_steps, n_barriers, shape, n_after in res:
        assert n_exchange == 1 and n_steps == 0 and n_barriers == 1 and n_after == 1 and shape == (14, 768)
        assert gather_bytes == 7 * 768 * 4                  # gloo path: a list all_gather of (7, 768) fp32 shards (the NCCL path gathers into one tensor)


def test_roofline_fractions_are_against_the_hardware_peak():
    """every MFMA roofline block prices `frac` against the 2500 TFLOP/s hardware peak; the fp32 mode's derating by issued MFMA units is a
    separate field (`frac_of_issue_bound` = frac x units), never the peak."""
    import bench
    r = bench.roofline_mfma("flash_attn_kernel", 1068.0, None, {"traffic": None})
    assert r["peak"] == 2500.0 and r["frac"] == round(1068.0 / 2500.0, 4) and "frac_of_issue_bound" not in r
    f = bench.roofline_mfma("flash_attn_split_kernel", 436.0, 2.5, {"traffic": 123})
    assert f["peak"] == 2500.0 and f["frac"] == round(436.0 / 2500.0, 4) == 0.1744
    assert f["issued_units_per_product"] == 2.5 and f["frac_of_issue_bound"] == round(436.0 * 2.5 / 2500.0, 4)
    assert f["exact_fp32_matrix_peak"] == 157.3 and f["traffic"] == 123
