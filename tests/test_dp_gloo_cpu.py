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


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` exactly as the driver calls it (no torch.distributed.run around it, WORLD_SIZE unset):
    the parent starts two ranks before touching any GPU, they rendezvous (gloo here: no GPU), rank 0 prints ONE JSON
    line and the parent exits with the ranks' status.  --check-launch stops after the rendezvous (the step itself needs
    a GPU: tests/test_bench_gpu.py::test_bench_two_ranks_one_card)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WFAE_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--check-launch"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dp"]["ranks_seen"] == 2 and d["dp"]["backend"] == "gloo"


def test_bench_launcher_propagates_a_rank_failure():
    """a rank that dies must not leave the others waiting in a collective: the parent ends them and returns non-zero"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WFAE_DIST_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--check-launch"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
