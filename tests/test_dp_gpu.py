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
    recon, _ = net(x)
    Fn.l1_loss(recon, x).backward()
    Fn.join_side_stream()
    torch.cuda.synchronize()
    g_local = a.flat_g.clone()
    dp.reduce_gradients()
    g_sum = a.flat_g.clone()
    gl = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(gl, g_local)
    ok_sum = torch.allclose(g_sum, gl[0] + gl[1], rtol=0, atol=0)
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
