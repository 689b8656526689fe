"""Launch census of one AE train step: entry-point calls per step and serialised time at a small (dispatch-bound) size.

    python tools/launch_count.py [--batch 8] [--size 128]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import functional as Fn, ops, synth  # noqa: E402
from weatherforecastingtoolkit_amd.optim import FusedAdamW  # noqa: E402
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PosAwareAE_TF(img_size=a.size).to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=5e-5, weight_decay=1e-4)
    x = torch.from_numpy(synth.uniform_frames(a.batch, a.size, seed=1234)).to(dev)

    def step():
        recon, _ = net(x)
        loss = Fn.l1_loss(recon, x)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    Fn.set_wgrad_overlap(False)
    ops.profile_start()
    step()
    prof = ops.profile_stop()
    rows = sorted(prof.items(), key=lambda kv: -kv[1][1])
    print(f"{sum(v[0] for v in prof.values())} entry-point calls, {sum(v[1] for v in prof.values()):.2f} ms serialised")
    for k, v in rows:
        print(f"  {k:32s} {v[0]:5d} calls {v[1]:8.3f} ms  {1e3 * v[1] / v[0]:7.1f} us/call")


if __name__ == "__main__":
    main()
