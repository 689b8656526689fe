"""Feasibility probe: capture the whole AE train step (fwd + L1 + bwd + AdamW) in one HIP graph and compare
ms/step with the eager launch path at small batch (launch-bound) sizes.

    python tools/graph_probe.py [--batch 8] [--size 128] [--steps 30]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import functional as Fn, synth  # noqa: E402
from weatherforecastingtoolkit_amd.optim import FusedAdamW  # noqa: E402
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PosAwareAE_TF(img_size=a.size).to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=5e-5, weight_decay=1e-4)
    x = torch.from_numpy(synth.uniform_frames(a.batch, a.size, seed=1234)).to(dev)
    out = {}

    def step():
        recon, _ = net(x)
        loss = Fn.l1_loss(recon, x)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        out["loss"] = loss.detach()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    for _ in range(3):
        step()
    eager = timed(step, a.steps)
    print(f"eager: {eager:.2f} ms/step  loss {float(out['loss']):.6f}", flush=True)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(f"replay loss {float(out['loss']):.6f}", flush=True)
    graph = timed(g.replay, a.steps)
    print(f"graph: {graph:.2f} ms/step  loss {float(out['loss']):.6f}  speedup {eager / graph:.2f}x", flush=True)


if __name__ == "__main__":
    main()
