"""CPU, world_size 2 over gloo: the N>1 path of bench.py (stream partition + barrier + MAX-reduced time)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tools import stream_shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = stream_shard.assign_streams(7, world, rank)
    dist.barrier()
    elapsed = 1.0 + rank  # rank 1 is the slow one
    mx = stream_shard.max_over_ranks(dist, elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, mx, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_partition_and_time_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert all(r[2] == 2.0 for r in res)                       # MAX over ranks
    allstreams = sorted(s for part in res[0][3] for s in part)
    assert allstreams == list(range(7))                         # every stream exactly once, no overlap
    assert stream_shard.aggregate_throughput(30.0 * 8, world, 2.0) == 240.0


def test_assign_streams_edges():
    assert stream_shard.assign_streams(0, 4, 1) == []
    assert stream_shard.assign_streams(3, 8, 5) == []
    assert stream_shard.assign_streams(64, 8, 7) == list(range(7, 64, 8))
    import pytest
    with pytest.raises(ValueError):
        stream_shard.assign_streams(4, 2, 2)


def test_bench_launches_its_own_ranks():
    # `python bench.py --gpus 2` run plainly (no torch.distributed.run around it) starts the two ranks itself; --dry-run
    # keeps the rendezvous (gloo), the barrier, the MAX-reduced time and the single JSON line, and skips the GPU work.
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "streams",
                          "--streams", "7"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True
    assert rec["streams_per_rank"] == [4, 3]                 # stream i -> rank i mod 2
    assert rec["elapsed_max_s"] >= 0.02                      # rank 1 is the slow one (MAX over ranks)


def test_dry_run_line_of_the_default_workload_carries_the_streams_object():
    # `bench.py --gpus N` as the driver runs it (default workload = the weak-scaling headline) also measures BASELINE
    # configs[3] — 64 turbo streams sharded i mod N, strong scaling — on the same ranks and attaches it as `streams`
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["scaling"] == "weak" and rec["streams_per_rank"] == [1, 1]
    st = rec["streams"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 2 and st["streams_per_rank"] == [32, 32]
