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


def test_bench_launcher_stops_its_ranks_when_signalled():
    """a harness that kills the PARENT (`timeout`, a closed terminal) must not leave ranks behind holding a GPU or
    waiting in a collective: SIGTERM to `python bench.py --gpus 2` takes every rank down (own sessions, killpg) and the
    parent exits 128 + 15"""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WFAE_DIST_BACKEND="gloo", WFAE_BENCH_TEST_SLEEP="120", WFAE_BENCH_VERBOSE="1")
    p = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--check-launch"], cwd=root, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = ""
    for _ in range(20):                             # rank 0's JSON line (behind gloo's own chatter): both ranks are up and now asleep
        line = p.stdout.readline()
        if line.startswith("{") or not line:
            break
    assert line.startswith("{"), line
    p.send_signal(signal.SIGTERM)
    try:
        out, err = p.communicate(timeout=60)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-1000:])
    pids = [int(l.split()[-1]) for l in err.splitlines() if l.startswith("bench.py: rank ")]
    assert len(pids) == 2, err
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, f"rank process {pid} survived the parent"


def _reduce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from weatherforecastingtoolkit_amd import parallel
    from weatherforecastingtoolkit_amd.optim import FlatArena
    parallel.init_from_env("gloo")
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(300, 400), torch.nn.Linear(400, 10))
    params = list(model.parameters())

    class Opt:                                       # what DataParallelTrainer needs of FusedAdamW
        arenas = [FlatArena(params)]
        grad_scale = 1.0

    dp = parallel.DataParallelTrainer(model, Opt(), bucket_mb=1, overlap=False)   # 1 MiB buckets -> two all-reduces
    a = Opt.arenas[0]
    ok = True
    for mode in ("blocking", "split"):
        for i, p in enumerate(params):
            p.grad = None
            if i == 2:                               # one gradient OUTSIDE the arena (autograd summed two uses): exchanged by tensor
                p.grad = torch.full_like(p, float(rank + 1))
            else:
                p._wfae_grad_view.fill_(float((rank + 1) * (i + 1)))
                p.grad = p._wfae_grad_view.view(p.shape)
        if mode == "blocking":
            dp.reduce_gradients()
        else:
            dp.start_reduce()
            busy = torch.randn(200, 200) @ torch.randn(200, 200)      # work queued while the exchange is in flight
            dp.finish_reduce()
            ok = ok and bool(torch.isfinite(busy).all())
        for i, p in enumerate(params):
            want = 3.0 if i == 2 else 3.0 * (i + 1)
            ok = ok and bool((p.grad == want).all())
    try:
        dp.finish_reduce()
        ok = False
    except RuntimeError:
        pass
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_start_finish_reduce_world2_gloo():
    """the split exchange the AE+GAN step uses to put the generator's all-reduce under the discriminator's forward /
    backward (experiments/ae_v2_2/train.py::training_step): start_reduce() + finish_reduce() give the sums of
    reduce_gradients(), arena buckets and stray gradients alike; finish without start raises"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
