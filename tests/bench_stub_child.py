"""Stand-in for a bench.py rank (tests/test_bench_launcher.py): no torch, no GPU.  Prints what the launcher put in its environment;
rank 0 prints the ONE JSON line.  RZ_STUB_FAIL=R makes rank R exit 3; RZ_STUB_HANG=R makes rank R sleep (it must be terminated); RZ_STUB_HANG=all: every rank."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
args = sys.argv[1:]
print(f"stub rank {rank}/{world} local {os.environ['LOCAL_RANK']} port {os.environ['MASTER_PORT']} addr {os.environ['MASTER_ADDR']} args {args}", flush=True)
if os.environ.get("RZ_STUB_FAIL") == str(rank):
    sys.exit(3)
if os.environ.get("RZ_STUB_HANG") in (str(rank), "all"):
    time.sleep(600)
if rank == 0:
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "argv": args, "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                      "launched_by": os.environ.get("RZ_BENCH_LAUNCHED_BY")}), flush=True)
