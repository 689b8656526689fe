"""bench.py — SEVIR 384x384 frames/sec of the conv-AE train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no launcher starts its own N ranks (child processes, before this process
touches the GPU) and relays rank 0's line; under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

One "step" = forward + L1 loss + backward + (gradient all-reduce) + AdamW + LR
schedule of ae_64x8x8_lin.PosAwareAE_TF(img_size=384) on a batch of 32
synthetic SEVIR-shaped frames per GPU (BASELINE.json configs[1]; weak scaling:
global batch 32*N).  Frames are generated on the device before the timed
region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      the dominant kernel of the step, chosen from THIS run: every launch of
                every entry point is timed with an event pair on its launch stream (a
                second pass of the same K steps right after the timed region, streams
                serialised so concurrent launches do not inflate each other; entry
                points that launch two kernels are timed per kernel) and the label with
                the largest total wins (two labels within 3 % of each other are a tie in one run's
                timings: then the one that tops the committed rocprofv3 kernel-stats summary of this
                command, profiles/dominant_kernel.json, is reported and the other is `runner_up`);
                bound = the larger of the label's summed MFMA and HBM terms;
                achieved = algorithmic bytes (or FLOPs) / time.
                `traffic` (HBM bytes per launch) comes from the committed PMC pass of
                this same command (profiles/pmc_traffic.json) and carries its source
                tag; it is reported as null + "stale" when the kernel sources have
                changed since that pass.
                Entry points that launch the same kernel are added up first (the split-operand
                Winograd GEMMs: wfae_wino_gemm_down + _up = sgemm3_kernel<B = K x N>), as rocprofv3's
                per-kernel statistics do; for that kernel `achieved` counts the executed bf16 MFMA
                work (six products per fp32 product) against the dense bf16 peak and
                `fp32_equivalent_tflops` the same work counted once.
  bf16_storage  the same step timed in the same run at `--precision medium` (bf16 activation storage + bf16 MFMA operands:
                BASELINE config 5's regime): {ms_per_step, value, peak_mem_GiB, hbm_fraction} — a side figure, never `value`.
  fp32_mfma_only  the same step timed in the same run with WFAE_SPLIT_GEMM=0 (every GEMM on
                v_mfma_f32_32x32x2_f32): a side figure, never `value`; config.matmul says which
                GEMMs the headline runs on the bf16 pipe with exact three-plane operands.
  step_roofline the whole step against its roofs:
                ideal_ms = sum over every launch of max(algorithmic bytes / 8 TB/s, algorithmic FLOPs / peak of the
                matrix instruction THAT launch runs on — 157.3 TF for v_mfma_f32_32x32x2_f32, 2500 / 6 TF fp32-equivalent
                for the exact three-plane bf16 split, 2500 TF for bf16 operands), achieved = ideal_ms / ms_per_step;
                hbm_fraction = SURVEY.md 8(d)'s fused-minimum bytes per frame (7.365 GB at fp32, half of it with bf16
                activation storage) * fps / 8 TB/s; hbm_bytes_per_step_measured = the PMC byte count of the committed
                rocprofv3 pass of this command (profiles/pmc_traffic.json; null + "stale" when the kernel sources changed
                since) — the excess over the algorithmic bytes is re-read / unfused traffic.
  compute_path  which arithmetic `value` was measured on (dtype stays the tensor / accumulation type).
  cpu_baseline  the oracle (CPU restatement of the reference path, kind "port")
                timed on this box's host cores on a bounded sample: one warm-up + one
                timed full train step at 384x384, batch 4 = BASELINE configs[0]
                (rank 0, N=1 only).
  dp            (N > 1) ranks seen by an all-reduce of ones and the event-timed
                gradient exchange per step.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOPS_PER_FRAME_384 = 938.8e9   # fwd+bwd, SURVEY.md §8(d)
BYTES_PER_FRAME_384 = 7.365e9   # fused-minimum HBM traffic, SURVEY.md §8(d)
PEAK_FP32 = 157.3e12            # MI355X_MICROARCH.md: fp32 vector = fp32 matrix
PEAK_HBM = 8.0e12

PEAK_BF16 = 2500.0e12           # dense bf16 MFMA (MI355X_MICROARCH.md)

# labels whose launches run ONE kernel are ranked together (rocprofv3's kernel-trace stats rank by kernel): with the split
# operands the down and up products of the Winograd layers are the same sgemm3_kernel<B = K x N>
KERNEL_GROUP_SPLIT = {"wfae_wino_gemm_down": "split_gemm[B=KxN]", "wfae_wino_gemm_up": "split_gemm[B=KxN]",
                      "wfae_wino_gemm_wgrad": "split_gemm[B=NxK]"}

KERNEL_OF = {
    "split_gemm[B=KxN]": "sgemm3_kernel<0> (Winograd-domain GEMMs M_xi = U_xi V_xi and dV_xi = U_xi^T Mt_xi of the 4x4 s2 "
                         "layers, 25 per launch: fp32 operands as three bf16 planes, six v_mfma_f32_16x16x32_bf16 products "
                         "per fp32 product (one plane, one product at 'medium'), fp32 accumulation — csrc/splitgemm.hip)",
    "split_gemm[B=NxK]": "sgemm3_kernel<1> (Winograd-domain weight-gradient GEMMs dU_xi = Mt_xi V_xi^T, split-K, the same "
                         "split-operand bf16 MFMA scheme)",
    "wfae_bn_act_bwd[dx]": "bn_act_bwd_dx_kernel<GELU> (BatchNorm + GELU backward: dx from dy, x (+ residual-branch gradient), "
                           "HBM-bound streaming kernel)",
    "wfae_bn_act_bwd[reduce]": "bn_act_bwd_reduce_kernel<GELU> (per-channel sums of dU and dU*xhat, fp64 accumulation)",
    "wfae_wino_gemm_wgrad": "gemm_kernel<256,2,2,A_KCONTIG,B_KCONTIG,E_SLAB> (Winograd-domain weight-gradient GEMMs "
                            "dU_xi = Mt_xi V_xi^T of the 4x4 s2 convs, 9 per launch, split-K, fp32 MFMA)",
    "wfae_wino_gemm_down": "gemm_kernel<256,2,2,A_KCONTIG,B_NCONTIG,E_BATCHED> (Winograd-domain GEMMs M_xi = U_xi V_xi, "
                           "Conv2d fwd / ConvTranspose2d dgrad, 9 per launch, fp32 MFMA)",
    "wfae_wino_gemm_up": "gemm_kernel<256,2,2,A_MCONTIG,B_NCONTIG,E_BATCHED> (Winograd-domain GEMMs dV_xi = U_xi^T Mt_xi, "
                         "ConvTranspose2d fwd / Conv2d dgrad, 9 per launch, fp32 MFMA)",
    "wfae_conv4x4s2_up": "gemm_kernel<B_UP> (ConvTranspose2d fwd / Conv2d dgrad, fp32 MFMA implicit GEMM)",
    "wfae_conv4x4s2_down": "gemm_kernel<B_DOWN> (Conv2d fwd / ConvTranspose2d dgrad, fp32 MFMA implicit GEMM)",
    "wfae_conv4x4s2_wgrad": "gemm_kernel<B_WGRAD> (4x4 s2 weight gradient, fp32 MFMA implicit GEMM, split-K)",
    "wfae_c1b_fwd": "c1b_kernel (1x1 convolutions of the Bottleneck on bf16 tensors, forward: 16-byte bf16 loads straight into the "
                    "MFMA LDS image, BatchNorm + GELU prologue or residual + BatchNorm-sum epilogue — csrc/c1b.hip)",
    "wfae_c1b_dgrad": "c1b_kernel (the same kernel with the transposed weight image: 1x1 data gradients on bf16 tensors)",
    "wfae_conv1x1_fwd": "c1r_kernel (1x1 conv fwd, register-direct, exact three-plane bf16 operands — csrc/c1r.hip) at the shapes it "
                        "serves, else gemm_kernel<A_KCONTIG,B_NCONTIG> (fp32 MFMA)",
    "wfae_conv1x1_bwd_data": "c1r_kernel (1x1 conv data gradient; at C <= 256 the widening one leaves only the sums of the "
                             "BatchNorm backward in front: wfae_c1r_bnred without a store) or gemm_kernel<A_MCONTIG,B_NCONTIG>",
    "wfae_c1r_bndx": "c1r_kernel<BNM = 3> (the widening 1x1 data gradient of a C <= 256 Bottleneck computed again with the second "
                     "pass of its first BatchNorm's backward in the epilogue: reads dT, x, skip gradient, writes dx — csrc/c1r.hip)",
    "wfae_conv1x1_bwd_weight": "c1w_kernel (1x1 conv wgrad, both operands as K-contiguous LDS rows, split-K — csrc/c1w.hip)",
}


def cpu_baseline(img_size=384, batch=1):
    """Time the oracle's full train step on the host cores (bounded sample)."""
    import torch
    from oracle import ae_oracle as orc
    from weatherforecastingtoolkit_amd import synth
    # the 1-GPU box's CPU share is 16 cores even though the host shows 256 logical CPUs:
    # using them all only thrashes (measured: 379 s instead of ~25 s for this sample)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp, kind in synth.ae_state_dict_spec(img_size):
        if kind == "bn_n":
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif kind in ("bn_w", "bn_rv"):
            sd[k] = torch.ones(shp)
        elif kind in ("bn_b", "bn_rm"):
            sd[k] = torch.zeros(shp)
        elif kind == "normal":
            sd[k] = torch.randn(shp, generator=g)
        else:
            fan = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1) if len(shp) > 1 else kind[1]
            b = 1.0 / fan ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        if sd[k].dtype.is_floating_point and "running_" not in k:
            sd[k].requires_grad_(True)
    x = torch.from_numpy(synth.uniform_frames(batch, img_size, seed=99))
    opt = orc.make_optimizer([p for _, p in orc.trainable(sd)], lr=5e-5, weight_decay=1e-4)
    t0 = time.perf_counter()
    orc.train_step(x, sd, opt)          # warm-up: oneDNN primitive creation, allocator growth
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.train_step(x, sd, opt)
    dt = time.perf_counter() - t0
    return {"value": batch / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ae_oracle.py train step (fwd+L1+bwd+AdamW), {img_size}x{img_size}, batch {batch}, "
                      f"torch CPU fp32, {cores} threads: 1 warm-up step ({cold:.1f} s, cold) + 1 timed step ({dt:.1f} s, warm)"}


def kernel_source_tag():
    """sha256 over the kernel sources: profiles/pmc_traffic.json records the tag of the build it was measured on"""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "weatherforecastingtoolkit_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def launch_ranks(n, argv, deadline_s=None):
    """`python bench.py --gpus N` (N > 1) with no launcher around it: start the N ranks as CHILD processes, one per GPU
    (RANK = LOCAL_RANK = 0..N-1, rendezvous on 127.0.0.1), relay rank 0's JSON line, return the worst exit status.
    Runs before anything in this process has imported torch or touched HIP: a process that has initialised the GPU must
    neither fork ranks nor exec (the reference leaves this to Lightning's launcher, experiments/ae_v2/train.py:332-343).
    Every rank runs in a session of its own; SIGTERM / SIGINT / SIGHUP to this parent (a harness `timeout`, a closed terminal)
    and the overall deadline take all of them down — terminate, then kill, by process group — so no rank is left holding a GPU
    or waiting in a collective.  The rendezvous port is picked by binding port 0; if a rank fails to bind it (taken in
    between) the launch is retried on a fresh port."""
    import signal
    import socket
    import subprocess

    procs = []

    def stop_all(grace=3.0):
        for sig in (signal.SIGTERM, signal.SIGKILL):
            live = [p for p in procs if p.poll() is None]
            if not live:
                return
            for p in live:
                try:
                    os.killpg(p.pid, sig)          # start_new_session: the rank's pid is its process-group id
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.time() + grace
            while time.time() < t_end and any(p.poll() is None for p in live):
                time.sleep(0.05)

    class _Stop(Exception):
        pass

    def on_signal(signum, frame):
        raise _Stop(signum)

    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    deadline = time.time() + (deadline_s if deadline_s else float(os.environ.get("WFAE_BENCH_DEADLINE_S", "3000")))
    rc = 0
    try:
        for attempt in range(3):
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n))
            base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between processes on this driver
            base.setdefault("OMP_NUM_THREADS", "4")
            del procs[:]
            for r in range(n):
                env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
                # rank 0 owns stdout (the ONE JSON line); the other ranks' stdout goes to stderr so nothing else lands on it
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                              stdout=None if r == 0 else sys.stderr, start_new_session=True))
                if os.environ.get("WFAE_BENCH_VERBOSE"):
                    print(f"bench.py: rank {r} pid {procs[-1].pid}", file=sys.stderr, flush=True)
            rc = 0
            pending = set(range(n))
            while pending:
                if time.time() > deadline:
                    print(f"bench.py: {n}-rank run exceeded its deadline; stopping every rank", file=sys.stderr)
                    rc = 124
                    break
                for r in list(pending):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    pending.discard(r)
                    if code != 0:
                        rc = rc or code
                        pending.clear()             # a rank died: the others would wait in a collective for ever
                        break
                time.sleep(0.05)
            stop_all()
            if rc != EADDRINUSE_RC:
                break
    except _Stop as e:
        print(f"bench.py: signal {e.args[0]}: stopping every rank", file=sys.stderr)
        rc = 128 + int(e.args[0])
    finally:
        stop_all()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    return rc


EADDRINUSE_RC = 98   # a rank's exit status when the rendezvous port was taken between the parent's probe and its bind


def check_launch(args):
    """--check-launch: the rendezvous of a multi-rank run without the model (CPU test of the self-launcher)"""
    import torch
    import torch.distributed as dist
    from weatherforecastingtoolkit_amd import parallel
    backend = os.environ.get("WFAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    rank, world, local = parallel.init_from_env(backend)
    ones = torch.ones(1, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(ones)
    if rank == 0:
        print(json.dumps({"check_launch": True, "n_gpus": world, "requested": args.gpus,
                          "dp": {"ranks_seen": int(ones.item()), "backend": backend if world > 1 else None}}), flush=True)
    if os.environ.get("WFAE_BENCH_TEST_SLEEP"):      # tests/test_dp_gloo_cpu.py: keep the ranks alive so the parent can be signalled
        time.sleep(float(os.environ["WFAE_BENCH_TEST_SLEEP"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU")
    ap.add_argument("--img-size", type=int, default=384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run weight gradients on the main stream")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the extra fp32-MFMA-only timing (profiling runs)")
    ap.add_argument("--no-medium-leg", action="store_true", help="skip the extra bf16-storage ('medium') timing")
    ap.add_argument("--fp32-mfma-only", action="store_true",
                    help="keep the Winograd-domain GEMMs on v_mfma_f32_32x32x2_f32 (no split bf16 operands)")
    ap.add_argument("--check-launch", action="store_true",
                    help="rendezvous only: every rank joins the process group, an all-reduce of ones counts them, rank 0 "
                         "prints {n_gpus, dp.ranks_seen} and exits (no model, no kernels; gloo when there is no GPU)")
    ap.add_argument("--precision", choices=["highest", "medium"], default="highest",
                    help="'medium' = BASELINE config 5's regime: bf16 MFMA operands AND bf16 activation storage in HBM "
                         "(NOT the headline configuration — the line is then labelled dtype bf16)")
    ap.add_argument("--fp32-tensors", action="store_true",
                    help="with --precision medium: keep the activation tensors fp32 in HBM (round 2's 'medium')")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process only starts the ranks (it never touches the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    if args.check_launch:
        return check_launch(args)
    from weatherforecastingtoolkit_amd import functional as Fn, ops, parallel
    from weatherforecastingtoolkit_amd.optim import CosineWarmupLR, FusedAdamW
    from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # WFAE_DIST_BACKEND=gloo lets several ranks share one card (rehearsal on a 1-GPU box); default RCCL
    backend = os.environ.get("WFAE_DIST_BACKEND", "nccl")
    if backend != "nccl":
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    want = int(os.environ.get("WORLD_SIZE", "1"))
    if backend == "nccl" and want > 1 and torch.cuda.device_count() < want:
        # RCCL needs one device per rank: say so before any rank enters a collective it could never leave
        print(f"bench.py: --gpus {want} over RCCL needs {want} visible GPUs, this node shows {torch.cuda.device_count()} "
              "(WFAE_DIST_BACKEND=gloo rehearses several ranks on one card)", file=sys.stderr)
        sys.exit(2)
    try:
        rank, world, local = parallel.init_from_env(backend)
    except RuntimeError as e:      # the rendezvous port was taken between the parent's probe and rank 0's bind: the parent retries
        if "address already in use" in str(e).lower() or "EADDRINUSE" in str(e):
            sys.exit(EADDRINUSE_RC)
        raise
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    Fn.set_wgrad_overlap(not args.no_overlap)
    ops.set_float32_matmul_precision(args.precision)
    if args.precision == "medium" and args.fp32_tensors:
        ops.set_activation_storage(torch.float32)
    bf16_storage = ops.activation_dtype() == torch.bfloat16
    if args.fp32_mfma_only:
        ops.set_split_gemm(False)
    torch.manual_seed(0)  # identical random-init weights on every rank (then broadcast anyway)
    net = PosAwareAE_TF(img_size=args.img_size).to(dev).train()
    opt = FusedAdamW(net.parameters(), lr=5e-5, betas=(0.9, 0.999), weight_decay=1e-4)
    total_steps = max(args.steps + args.warmup, 10)
    sched = CosineWarmupLR(opt, 5e-6, 5e-7, 5e-5, total_steps, 0.1 * total_steps)
    dp = parallel.DataParallelTrainer(net, opt)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    B, S = args.batch, args.img_size
    x = (torch.randint(0, 256, (B, 1, S, S), generator=gen, device=dev, dtype=torch.int32).float() / 255.0).contiguous()

    xchg = []   # (start, end) event pairs around the gradient exchange, filled during the read-out pass

    def step(time_exchange=False):
        opt.zero_grad(set_to_none=True)
        recon, _ = net(x)
        loss = Fn.l1_loss(recon, x)
        loss.backward()
        if time_exchange and world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            Fn.join_side_stream()
            e0.record()
            dp.reduce_gradients()
            e1.record()
            xchg.append((e0, e1))
        else:
            dp.reduce_gradients()
        opt.step()
        sched.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    # Per-kernel read-out for the roofline object: the same step, K more times, with an event pair
    # around every launch on its launch stream.  Streams are serialised for this pass (weight
    # gradients back on the main stream) so a launch's duration is not inflated by a concurrent one.
    prof = {}
    if not args.no_kernel_timing:
        Fn.set_wgrad_overlap(False)
        step()
        fence()
        ops.profile_start()
        for _ in range(args.steps):
            step(time_exchange=True)
        prof = ops.profile_stop()
        Fn.set_wgrad_overlap(not args.no_overlap)
    # the same step with every GEMM on the fp32 MFMA instruction (reported beside the headline, never as `value`)
    strict = None
    if args.precision == "highest" and ops.split_gemm_enabled() and not args.no_fp32_leg:
        ops.set_split_gemm(False)
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        strict = (time.perf_counter() - t1) / args.steps
        ops.set_split_gemm(True)
    # BASELINE config 5's precision in the same run: bf16 activation storage + bf16 MFMA operands ('medium'), same model,
    # optimiser state, batch, steps and warm-up rule — a side figure beside the headline, never `value`
    peak_mem_headline = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    medium = None
    if args.precision == "highest" and not args.no_medium_leg:
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        ops.set_float32_matmul_precision("medium")
        ops.set_activation_storage(torch.bfloat16)
        try:
            for _ in range(max(1, args.warmup)):
                step()
            fence()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            medium = ((time.perf_counter() - t2) / args.steps, torch.cuda.max_memory_allocated(dev) / 2 ** 30)
        finally:
            ops.set_float32_matmul_precision("highest")
        if world > 1:
            tm = torch.tensor([medium[0]], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            medium = (float(tm.item()), medium[1])
    ranks_seen, xchg_ms = 1, None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        if xchg:
            torch.cuda.synchronize()
            xchg_ms = sum(a.elapsed_time(b) for a, b in xchg) / len(xchg)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    final_loss = float(loss.item())

    if rank == 0:
        fps = world * B * args.steps / dt
        scale = (S / 384.0) ** 2
        out = {
            "metric": "SEVIR 384x384 frames/sec (AE train step)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "highest" else "bf16", "data": "synthetic",
            "compute_path": ("fp32 tensors + fp32 accumulation; MFMA-bound GEMMs as exact bf16x3 splits on "
                             "v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16 (6 products per fp32 product), the rest on v_mfma_f32_32x32x2_f32"
                             if args.precision == "highest" and ops.split_gemm_enabled() else
                             "fp32 tensors, every GEMM on v_mfma_f32_32x32x2_f32" if args.precision == "highest" else
                             "bf16 activation storage + bf16 MFMA operands, fp32 accumulation / parameters / statistics"
                             if bf16_storage else "fp32 tensors, bf16 MFMA operands, fp32 accumulation"),
            "config": {"workload": f"experiments/ae_v2 conv AE (ae_64x8x8_lin.PosAwareAE_TF, img_size={S}), "
                                   f"1x{S}x{S} synthetic SEVIR frames, batch {B}/GPU, "
                                   f"{'fp32' if args.precision == 'highest' else ('bf16 activations' if bf16_storage else 'bf16 MFMA operands / fp32 tensors')}, "
                                   "fwd + L1 + bwd + AdamW + cosine-warmup LR",
                       "batch_per_gpu": B, "global_batch": B * world, "img_size": S,
                       "parallelism": f"dp{world}", "params": sum(p.numel() for p in net.parameters()),
                       "matmul": ("fp32 tensors, fp32 accumulation; the MFMA-bound GEMMs (Winograd-domain products of the 4x4 "
                                  "stride-2 layers, 1x1 convolutions with K >= 128) multiply on the bf16 matrix pipe with "
                                  "every fp32 operand carried EXACTLY as three bf16 values (six products per fp32 product: "
                                  "error vs fp64 that of the fp32 MFMA kernels, tests/test_split_gemm_gpu.py); the other "
                                  "GEMMs on v_mfma_f32_32x32x2_f32; see fp32_mfma_only for the step without the split"
                                  if args.precision == "highest" and ops.split_gemm_enabled() else
                                  "v_mfma_f32_32x32x2_f32 everywhere" if args.precision == "highest" else
                                  "bf16-rounded MFMA operands")},
            "final_loss": final_loss,
            "peak_mem_GiB": peak_mem_headline,
            "step_roofline": {"hbm_fraction": BYTES_PER_FRAME_384 * (0.5 if bf16_storage else 1.0) * scale * fps / world / PEAK_HBM,
                              "algorithmic_bytes_per_frame": BYTES_PER_FRAME_384 * (0.5 if bf16_storage else 1.0) * scale,
                              "direct_form_gflop_per_frame": FLOPS_PER_FRAME_384 * scale / 1e9},
        }
        if strict is not None:
            out["fp32_mfma_only"] = {"ms_per_step": 1e3 * strict, "value": world * B / strict, "unit": "frames/s",
                                     "note": "same step, same run, WFAE_SPLIT_GEMM=0: every GEMM on v_mfma_f32_32x32x2_f32"}
        if medium is not None:
            m_fps = world * B / medium[0]
            out["bf16_storage"] = {"ms_per_step": 1e3 * medium[0], "value": m_fps, "unit": "frames/s", "peak_mem_GiB": medium[1],
                                   "hbm_fraction": 0.5 * BYTES_PER_FRAME_384 * scale * m_fps / world / PEAK_HBM,
                                   "note": "same step, same run at --precision medium (BASELINE config 5's regime, "
                                           "experiments/ae_v2_2/train.py:223): bf16 activation storage + bf16 MFMA operands, fp32 "
                                           "accumulation / parameters / statistics; a side figure, never `value`"}
        if world > 1:
            out["dp"] = {"ranks_seen": ranks_seen, "backend": backend, "overlap": bool(dp._hooks),
                         "nccl_algo": os.environ.get("NCCL_ALGO"),
                         "grad_exchange_ms_per_step": xchg_ms,
                         "payload_bytes": 4 * sum(a.numel for a in opt.arenas)}
        if prof:
            tot_ms = sum(v[1] for v in prof.values())
            fam = sorted(prof.items(), key=lambda kv: -kv[1][1])
            # the dominant kernel of THIS run: the kernel with the largest total event-timed duration (entry points
            # that launch two kernels are timed per kernel through ops' `phases` labels; entry points that share one
            # kernel are added up)
            split_on = ops.split_gemm_enabled()
            grp = {}
            for k, v in prof.items():
                gk = KERNEL_GROUP_SPLIT.get(k, k) if split_on else k
                g0 = grp.setdefault(gk, [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
                for i in range(7):
                    g0[i] += v[i]
            ranked = sorted(grp.items(), key=lambda kv: -kv[1][1])
            pj = {}
            try:   # the committed PMC / kernel-stats summaries of this same command (profiles/)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pj = json.load(f)
            except (OSError, ValueError):
                pass
            # two kernels within 3 % of each other are a tie in this run's timings (run-to-run spread is larger): the one that
            # tops the committed rocprofv3 kernel-stats summary of this command (profiles/dominant_kernel.json) is reported
            # and the other one is named beside it
            pick = 0
            if len(ranked) > 1 and ranked[1][1][1] >= 0.97 * ranked[0][1][1]:
                try:
                    with open(os.path.join(ROOT, "profiles", "dominant_kernel.json")) as f:
                        if json.load(f).get("entry_point") == ranked[1][0]:
                            pick = 1
                except (OSError, ValueError):
                    pass
            exec_flops_step = sum(v[2] for v in prof.values()) / args.steps
            ideal_ms = sum(v[4] for v in prof.values()) / args.steps
            out["step_roofline"]["executed_gflop_per_frame"] = exec_flops_step / B / 1e9
            out["step_roofline"]["ideal_ms"] = ideal_ms
            out["step_roofline"]["achieved"] = ideal_ms / (1e3 * dt / args.steps)
            out["step_roofline"]["note"] = ("ideal_ms = sum over launches of max(algorithmic bytes / 8 TB/s, algorithmic FLOPs / "
                                            "peak of the matrix instruction the launch runs on); achieved = ideal_ms / ms_per_step")
            tag = kernel_source_tag()
            stale = pj.get("_kernel_source_tag") != tag
            if pj:   # measured HBM bytes of the whole step
                key = "_step_total_bf16" if args.precision == "medium" else "_step_total"
                tot = pj.get(key)
                out["step_roofline"]["hbm_bytes_per_step_measured"] = None if (stale or not tot) else tot
                out["step_roofline"]["hbm_bytes_source"] = {"file": "profiles/pmc_traffic.json", "measured_on": pj.get("_source"),
                                                            "stale": stale, "key": key}
            nprod = 6 if args.precision == "highest" else 1   # bf16 MFMA products per product of the transform-domain GEMMs

            def roof(name, rec):
                """the roof that binds a kernel's launches as a whole: the larger of its summed MFMA and HBM terms"""
                calls, ms, fl, by, _, t_mfma, t_hbm = rec
                if t_mfma >= t_hbm and fl > 0:
                    bound, unit = "mfma", "TFLOP/s"
                    if name.startswith("split_gemm"):     # executed bf16 MFMA work against the dense bf16 peak
                        ach, peak = nprod * fl / (ms * 1e-3) / 1e12, PEAK_BF16 / 1e12
                    else:                                 # fp32-equivalent work against the (mean) peak of the instructions it ran on
                        ach, peak = fl / (ms * 1e-3) / 1e12, fl / (t_mfma * 1e-3) / 1e12
                else:
                    bound, unit, ach, peak = "hbm", "GB/s", by / (ms * 1e-3) / 1e9, PEAK_HBM / 1e9
                r = {"kernel": KERNEL_OF.get(name, name), "entry_point": name, "bound": bound, "achieved": ach, "peak": peak,
                     "unit": unit, "frac": ach / peak, "launches": calls, "avg_launch_ms": ms / calls,
                     "algorithmic_per_launch": ((nprod if name.startswith("split_gemm") else 1) * fl if bound == "mfma" else by) / calls,
                     "share_of_kernel_time": ms / tot_ms}
                if name.startswith("split_gemm"):
                    r["fp32_equivalent_tflops"] = fl / (ms * 1e-3) / 1e12
                    r["note"] = ("achieved = executed bf16 MFMA FLOPs (%d product%s per product of the Winograd-domain GEMMs) against "
                                 "the dense bf16 peak; fp32_equivalent_tflops = the same work counted once, comparable with the "
                                 "157.3 TF fp32 MFMA peak" % (nprod, "s" if nprod > 1 else ""))
                return r

            name, rec = ranked[pick]
            out["roofline"] = roof(name, rec)
            ent = pj.get(name, {})
            out["roofline"]["traffic"] = None if stale else ent.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = {"file": "profiles/pmc_traffic.json", "measured_on": pj.get("_source", "?"),
                                                 "kernel_source_tag": pj.get("_kernel_source_tag"), "this_build": tag, "stale": stale}
            if len(ranked) > 1:
                o = ranked[1 - pick] if pick else ranked[1]
                ru = roof(o[0], o[1])
                out["roofline"]["runner_up"] = {k: ru[k] for k in ("kernel", "entry_point", "bound", "achieved", "peak", "unit", "frac",
                                                                   "launches", "avg_launch_ms", "share_of_kernel_time")}
                out["roofline"]["runner_up"]["tie"] = bool(ranked[1][1][1] >= 0.97 * ranked[0][1][1])
            out["kernel_breakdown"] = [
                {"entry_point": k, "kernel": (KERNEL_GROUP_SPLIT.get(k, k) if split_on else k),
                 "calls_per_step": v[0] / args.steps, "ms_per_step": v[1] / args.steps,
                 "tflops": (v[2] / (v[1] * 1e-3) / 1e12) if v[2] else None,
                 "gbps": v[3] / (v[1] * 1e-3) / 1e9, "ideal_ms_per_step": v[4] / args.steps} for k, v in fam[:16]]
        if world == 1 and not args.no_cpu_baseline:
            del net, opt, dp
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.img_size, 4)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
