"""Path-B timings: AE_ViT_2048 train step (B frames of 128x128) and the linear forecaster step on its token latent.
    python tools/vit_bench.py [--batch 32]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import functional as Fn, ops, synth  # noqa: E402
from weatherforecastingtoolkit_amd.optim import FusedAdamW  # noqa: E402
from weatherforecastingtoolkit_amd.pipeline.models.ae_vit import AE_ViT_2048  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = AE_ViT_2048().to(dev).train()
opt = FusedAdamW(net.parameters(), lr=1e-4)
x = torch.from_numpy(synth.uniform_frames(a.batch, 128, seed=1)).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    out, z = net(x)
    loss = Fn.mse_loss(out, x)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 100
print(f"AE_ViT_2048 train step B={a.batch}: {ms:.1f} ms  ({a.batch / ms * 1e3:.0f} frames/s)")
ops.profile_start()
step()
prof = ops.profile_stop()
for k, (calls, tms, fl, by, *_) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"   {k:26s} {calls:4d} calls {tms:8.2f} ms  {fl / tms / 1e9 if tms else 0:7.1f} TF/s")
# forecaster on the [64, 512] token latent: 13 -> 12 frames, B sequences
from weatherforecastingtoolkit_amd import nn as wnn  # noqa: E402
B = 8
v = torch.randn(B, 25, 512, 8, 8, device=dev)
pred = wnn.Linear(13 * 512, 12 * 512).to(dev)
popt = FusedAdamW(pred.parameters(), lr=1e-4)


def fstep():
    popt.zero_grad(set_to_none=True)
    X, Y = ops.latent_diff_pack(v, 13)
    loss = Fn.mse_loss(pred(X), Y)
    loss.backward()
    popt.step()


for _ in range(3):
    fstep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    fstep()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 100
fl = 3 * 2 * B * 64 * (13 * 512) * (12 * 512)
print(f"linear forecaster step on token latents, {B} sequences: {ms:.2f} ms  ({fl / ms / 1e9:.1f} TF/s over fwd+wgrad+[dgrad skipped])")
