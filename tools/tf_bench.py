"""`_tf` model (what experiments/ae_v2/train.py imports) train step at its native 128x128 size: ms/step and the
entry points of the latent transformer, serialised.

    python tools/tf_bench.py [--batch 32]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import functional as Fn, ops, synth  # noqa: E402
from weatherforecastingtoolkit_amd.optim import FusedAdamW  # noqa: E402
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_tf import PosAwareAE_TF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PosAwareAE_TF().to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=5e-5, weight_decay=1e-4)
    x = torch.from_numpy(synth.uniform_frames(a.batch, 128, seed=1234)).to(dev)

    def step():
        recon, _ = net(x)
        loss = Fn.l1_loss(recon, x)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print(f"_tf AE train step B={a.batch} 128x128: {ms:.2f} ms  ({a.batch / ms * 1e3:.0f} frames/s)")
    Fn.set_wgrad_overlap(False)
    ops.profile_start()
    step()
    prof = ops.profile_stop()
    tf = {k: v for k, v in prof.items() if any(s in k for s in ("linear", "mha", "layernorm", "relu", "dropout", "add", "reduce_sum"))}
    print(f"transformer-side entry points: {sum(v[1] for v in tf.values()):.2f} ms of {sum(v[1] for v in prof.values()):.2f} ms serialised")
    for k, v in sorted(tf.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:30s} {v[0]:4d} calls {v[1]:7.3f} ms  {1e3 * v[1] / v[0]:6.1f} us/call")


if __name__ == "__main__":
    main()
