"""GPU: the data-parallel train step end to end with two ranks sharing cuda:0 over gloo
(the GPU box has one card; RCCL needs one device per rank, gloo moves the same buffers).
Checks the exchange bench.py / train.py perform: rank-0 parameter broadcast, sum all-reduce of
the flat gradient arena, 1/world mean inside AdamW, identical parameters afterwards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from weatherforecastingtoolkit_amd import functional as Fn, parallel, synth
    from weatherforecastingtoolkit_amd.optim import FusedAdamW
    from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF
    parallel.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.manual_seed(100 + rank)                      # different init per rank: the broadcast must fix it
    net = PosAwareAE_TF().to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=5e-5, weight_decay=1e-4)
    dp = parallel.DataParallelTrainer(net, opt, bucket_mb=64, overlap=True)
    assert len(dp._hooks) == 2, "backward hooks for the overlapped all-reduce were not installed"
    a = opt.arenas[0]
    p0 = a.flat_p.clone()
    Fn.set_wgrad_overlap(True)
    x = torch.from_numpy(synth.uniform_frames(4, 128, seed=7))[2 * rank:2 * rank + 2].to(dev)
    # the rank-local reference gradient comes from a backward with the hooks held back (no exchange in flight while
    # it is cloned); backward is bit-reproducible (test_fullsize_gpu.py), so the overlapped pass below produces the
    # same local values
    with dp.no_sync():
        recon, _ = net(x)
        Fn.l1_loss(recon, x).backward()
    Fn.join_side_stream()
    torch.cuda.synchronize()
    assert dp.hook_launches == 0 and dp._done_from is None and not dp._pending
    g_local = a.flat_g.clone()
    opt.zero_grad(set_to_none=True)
    recon, _ = net(x)
    Fn.l1_loss(recon, x).backward()
    hooks_fired = dp.hook_launches > 0 and dp._done_from is not None and len(dp._pending) > 0
    dp.reduce_gradients()
    torch.cuda.synchronize()
    g_sum = a.flat_g.clone()
    gl = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(gl, g_local)
    ok_sum = hooks_fired and torch.equal(g_sum, gl[0] + gl[1])
    opt.step()
    torch.cuda.synchronize()
    pl = [torch.zeros_like(a.flat_p) for _ in range(world)]
    dist.all_gather(pl, a.flat_p)
    p0l = [torch.zeros_like(p0) for _ in range(world)]
    dist.all_gather(p0l, p0)
    q.put((rank, bool(ok_sum), bool(torch.equal(pl[0], pl[1])), bool(torch.equal(p0l[0], p0l[1])),
           float(opt.grad_scale), float((pl[0] - p0l[0]).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_two_ranks_one_gpu(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok_sum, same_after, same_before, scale, moved in res:
        assert ok_sum, "all-reduced arena != sum of the ranks' gradients"
        assert same_before, "rank-0 parameters were not broadcast"
        assert same_after, "ranks diverged after the optimiser step"
        assert scale == 0.5 and 0 < moved < 1e-4


def _traj_worker(rank, world, port, q):
    """two optimiser steps, once with the hook-driven overlapped exchange and once with the post-backward exchange:
    the parameters must be bit-identical (same buckets, same sums; only the launch time differs)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    from weatherforecastingtoolkit_amd import functional as Fn, parallel, synth
    from weatherforecastingtoolkit_amd.optim import FusedAdamW
    from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_tf import PosAwareAE_TF
    parallel.init_from_env("gloo")
    dev = torch.device("cuda:0")
    x = torch.from_numpy(synth.uniform_frames(4, 128, seed=11))[2 * rank:2 * rank + 2].to(dev)
    out, unused_ok, fired = [], True, []
    for overlap in (False, True):
        torch.manual_seed(5)
        net = PosAwareAE_TF().to(dev).train()          # the _tf model: tf_encoder.* never receives a gradient
        opt = FusedAdamW(net.parameters(), lr=5e-5, weight_decay=1e-4)
        dp = parallel.DataParallelTrainer(net, opt, bucket_mb=64, overlap=overlap)
        Fn.set_wgrad_overlap(True)
        torch.manual_seed(99)                          # same counter-based dropout seeds in both runs
        Fn._seed_counter[0] = 0
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            recon, _ = net(x)
            Fn.l1_loss(recon, x).backward()
            unused_ok = unused_ok and dp.unused_slots_are_zero()
            dp.reduce_gradients()
            opt.step()
        torch.cuda.synchronize()
        fired.append(dp.hook_launches)
        out.append(opt.arenas[0].flat_p.clone())
    pl = [torch.zeros_like(out[1]) for _ in range(world)]
    dist.all_gather(pl, out[1])
    q.put((rank, bool(torch.equal(out[0], out[1])), bool(torch.equal(pl[0], pl[1])), bool(unused_ok), fired))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_overlap_matches_post_backward_exchange(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_traj_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for rank, same_traj, same_ranks, unused_ok, fired in res:
        assert fired[0] == 0 and fired[1] > 0, fired
        assert same_traj, "overlapped exchange changed the 2-step parameter trajectory"
        assert same_ranks, "ranks diverged"
        assert unused_ok, "gradient slots of never-used parameters are not all zero"


def _rccl_worker(port, q):
    """world_size 1 over the REAL backend: RCCL communicator creation, in-place all-reduce / broadcast on slices of the
    flat gradient arena (the exact calls GradSync issues), async handles as the backward hooks use them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        from weatherforecastingtoolkit_amd import parallel
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        sync = parallel.GradSync(bucket_mb=1)
        sync.world = 2                      # take the collective code path although the group has one rank
        flat = torch.arange(700_001, dtype=torch.float32, device="cuda:0")
        ref = flat.clone()
        sync.allreduce_(flat)               # sum over one rank: identity, through ncclAllReduce on 1 MiB buckets
        sync.broadcast_(flat, 0)
        hs = [dist.all_reduce(flat[o:o + 1000], async_op=True) for o in (0, 4096, 699_000)]   # offsets as the hooks use
        for h in hs:
            h.wait()
        ones = torch.ones(1, device="cuda:0")
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        q.put((bool(torch.equal(flat, ref)), int(ones.item()), dist.get_backend()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put(("error", repr(e), ""))


def test_rccl_backend_single_rank(dev):
    """The one-GPU box cannot hold two RCCL ranks (one device per rank); this at least drives the nccl (= RCCL) backend
    itself — communicator, bucketed in-place all-reduce, broadcast, async handles, barrier — on one rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=300)
    p.join(60)
    assert res[0] is True and res[1] == 1 and res[2] == "nccl", res
    assert p.exitcode == 0


def _gan_worker(rank, world, port, q):
    """AE+GAN step (experiments/ae_v2_2) on two ranks: the generator's all-reduce started before the discriminator's forward /
    backward and finished after it (BASELINE config 5's overlap) against the reference's literal serial order"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    try:
        import weatherforecastingtoolkit_amd.experiments.ae_v2_2 as pkg
        from weatherforecastingtoolkit_amd import config as C, functional as Fn, parallel, synth
        from weatherforecastingtoolkit_amd.experiments.ae_v2_2.train import CARRIED_KEYS, Model
        parallel.init_from_env("gloo")
        dev = torch.device("cuda:0")
        cfg = C.load(os.path.join(os.path.dirname(pkg.__file__), "config.yaml"), CARRIED_KEYS)
        cfg.trainer.total_train_steps = 10
        cfg.lpips.disc_start = 0
        x = torch.from_numpy(synth.uniform_frames(4, 128, seed=21))[2 * rank:2 * rank + 2].to(dev)
        out, started = [], []
        for overlap in (False, True):
            torch.manual_seed(7)
            model = Model(cfg, img_size=128).to(dev).train()
            model.overlap_exchange = overlap
            Fn.set_wgrad_overlap(True)
            model.configure_optimizers()
            calls = [0]
            orig = model._dp[0].start_reduce

            def counted(orig=orig, calls=calls):
                calls[0] += 1
                pending_before = len(model._dp[0]._pending)
                orig()
                calls.append(len(model._dp[0]._pending) - pending_before)

            model._dp[0].start_reduce = counted
            logs = None
            for _ in range(2):
                _, logs = model.training_step({"vil": x}, 0)
            Fn.join_side_stream()
            torch.cuda.synchronize()
            started.append(calls)
            out.append((model.g_opt.arenas[0].flat_p.clone(), model.d_opt.arenas[0].flat_p.clone(),
                        float(logs["train/g_grad_norm"]), float(logs["train/d_grad_norm"])))
            Fn.set_wgrad_overlap(False)
        same_order = torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and out[0][2:] == out[1][2:]
        gl = [torch.zeros_like(out[1][0]) for _ in range(world)]
        dist.all_gather(gl, out[1][0])
        dl = [torch.zeros_like(out[1][1]) for _ in range(world)]
        dist.all_gather(dl, out[1][1])
        same_ranks = torch.equal(gl[0], gl[1]) and torch.equal(dl[0], dl[1])
        q.put((rank, bool(same_order), bool(same_ranks), started))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", repr(e) + traceback.format_exc()[-1500:], None))


def test_gan_generator_exchange_under_discriminator_backward(dev):
    """VERDICT r3 item 3 / SURVEY 8(f) next-2: with the generator's gradient exchange overlapping the discriminator's forward
    and backward, two G-then-D optimiser steps give BIT-IDENTICAL generator and discriminator parameters and gradient norms
    to the serial order, on both ranks; the overlapped runs really took the split path (asynchronous all-reduces in flight
    when the discriminator's work was queued)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=900) for _ in ps)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for rank, same_order, same_ranks, started in res:
        assert same_order is True, (rank, same_order, same_ranks)
        assert same_ranks is True
        serial, over = started
        # serial order: start_reduce only from inside reduce_gradients (one per step); overlapped: called by training_step and
        # it left asynchronous work pending (> 0 handles) each time
        assert serial[0] == 2 and over[0] == 2 and all(n > 0 for n in over[1:]), started
