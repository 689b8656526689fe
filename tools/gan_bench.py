"""AE+GAN step (experiments/ae_v2_2) timing at the bench configuration: 384x384, fp32, B frames on one GPU.

    python tools/gan_bench.py [--batch 32] [--size 384] [--steps 5] [--warmup 2] [--no-gan]

Prints ms/step, frames/s and a per-entry-point breakdown (serialised second pass, events on the launch stream).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import config as C  # noqa: E402
from weatherforecastingtoolkit_amd import functional as Fn  # noqa: E402
from weatherforecastingtoolkit_amd import ops, synth  # noqa: E402
import weatherforecastingtoolkit_amd.experiments.ae_v2_2 as pkg  # noqa: E402
from weatherforecastingtoolkit_amd.experiments.ae_v2_2.train import CARRIED_KEYS, Model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-gan", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--serial-exchange", action="store_true", help="the reference's literal order: G exchange + step before the D pass")
    ap.add_argument("--precision", choices=["highest", "medium"], default="highest")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_float32_matmul_precision(a.precision)
    cfg = C.load(os.path.join(os.path.dirname(pkg.__file__), "config.yaml"), CARRIED_KEYS)
    cfg.trainer.total_train_steps = 1000
    cfg.lpips.disc_start = 10 ** 9 if a.no_gan else 0
    torch.manual_seed(0)
    model = Model(cfg, img_size=a.size).to(dev).train()
    Fn.set_wgrad_overlap(not a.no_overlap)
    model.configure_optimizers()
    model.overlap_exchange = not a.serial_exchange
    x = torch.from_numpy(synth.uniform_frames(a.batch, a.size, seed=1234)).to(dev)
    for _ in range(a.warmup):
        model.training_step({"vil": x}, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model.training_step({"vil": x}, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    print(json.dumps({"workload": f"ae_v2_2 AE{'' if a.no_gan else '+GAN'} step {a.size}x{a.size} B={a.batch} "
                                  f"{'fp32' if a.precision == 'highest' else 'bf16 operands'}",
                      "ms_per_step": ms, "frames_per_s": a.batch / ms * 1e3,
                      "mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "dp": {"world": model._dp[0].world, "overlap": bool(model.overlap_exchange),
                             "note": "generator all-reduce started before, finished after the discriminator's forward / backward "
                                     "(experiments/ae_v2_2/train.py::training_step); one rank here: no collective runs, never "
                                     "timed on RCCL"}}), flush=True)
    Fn.set_wgrad_overlap(False)
    ops.profile_start()
    for _ in range(2):
        model.training_step({"vil": x}, 0)
    prof = ops.profile_stop()
    rows = sorted(prof.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in prof.values()) / 2
    print(f"serialised kernel time {tot:.1f} ms/step")
    for k, (calls, tms, fl, by, *_) in rows[:24]:
        print(f"  {k:32s} {calls // 2:5d} calls {tms / 2:9.2f} ms  {fl / tms / 1e9 if tms else 0:7.1f} TF/s {by / tms / 1e6 if tms else 0:8.0f} GB/s")


if __name__ == "__main__":
    main()
