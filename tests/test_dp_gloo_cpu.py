"""CPU, world_size 2, gloo: the data-parallel exchange (bucketed sum all-reduce of
the flat gradient arena, rank-0 broadcast of parameters) that bench.py runs over
RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from weatherforecastingtoolkit_amd import parallel
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    sync = parallel.GradSync(bucket_mb=1)            # 1 MiB buckets -> several all-reduces
    n = 700_001
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    sync.allreduce_(g)
    ok1 = torch.equal(g, torch.arange(n, dtype=torch.float32) * 3)
    p = torch.full((1000,), float(rank + 5))
    sync.broadcast_(p, 0)
    ok2 = bool((p == 5).all())
    # mean via grad_scale: what FusedAdamW multiplies in
    ok3 = abs((g * (1.0 / w))[10].item() - 15.0) < 1e-6
    q.put((rank, ok1 and ok2 and ok3))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_is_identity():
    from weatherforecastingtoolkit_amd import parallel
    s = parallel.GradSync()
    t = torch.ones(10)
    assert s.world == 1 and s.allreduce_(t) is t and s.broadcast_(t) is t
