"""Per-layer kernel micro-benchmark at the real shapes of the 384x384, B=32 train step.
Times each libwfae.so entry point with events on the launch stream (interleaved
rounds, median) and prints ms / TFLOP/s / GB/s per shape.

    python tools/kbench.py [--only conv4,conv1,dconv,bn] [--batch 32] [--size 384] [--rounds 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, rounds):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def rnd(*shape):
    return torch.rand(shape, device=dev) - 0.5


def report(tag, ms, flops, nbytes):
    # ideal = the larger of the MFMA time at the LDS->MFMA structural ceiling (148 TF) and the HBM time at the
    # achievable streaming rate (6.3 TB/s), i.e. perfect overlap of the two pipes
    ideal = max(flops / 148e12, nbytes / 6.3e12) * 1e3
    print(f"{tag:58s} {ms:9.3f} ms  {flops / ms / 1e9:8.1f} TF/s  {nbytes / ms / 1e6:8.0f} GB/s  "
          f"ideal {ideal:7.3f} ms ({ideal / ms:4.0%})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="conv4,conv1,dconv,bn")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--precision", default="highest", choices=["highest", "medium"])
    a = ap.parse_args()
    only = set(a.only.split(","))
    ops.set_float32_matmul_precision(a.precision)
    B, S, R = a.batch, a.size, a.rounds
    tot = {}

    def acc(k, ms, mult=1):
        tot[k] = tot.get(k, 0.0) + ms * mult

    if "conv4" in only:
        for chi, clo, hlo in [(256, 512, S // 4), (512, 1024, S // 8), (1024, 1024, S // 16),
                              (512, 1024, S // 8), (256, 512, S // 4), (128, 256, S // 2)]:
            hi, lo, w = rnd(B, chi, 2 * hlo, 2 * hlo), rnd(B, clo, hlo, hlo), rnd(clo, chi, 4, 4)
            dw = torch.empty_like(w)
            fl = 32 * B * hlo * hlo * clo * chi
            by = 4 * (B * hlo * hlo * (clo + 4 * chi) + 16 * clo * chi)
            for nm, fn in [("down", lambda: ops.conv4x4s2_down(hi, w)), ("up", lambda: ops.conv4x4s2_up(lo, w)),
                           ("wgrad", lambda: ops.conv4x4s2_wgrad(lo, hi, dw))]:
                ms = timeit(fn, R)
                report(f"conv4x4 {nm:5s} hi {chi}@{2 * hlo} lo {clo}@{hlo}", ms, fl, by)
                acc("conv4_" + nm, ms)
            del hi, lo, w, dw
    if "wino" in only:
        # the Winograd pieces one by one, fp32 MFMA GEMMs against the split-operand bf16 GEMMs (splitgemm.hip)
        for chi, clo, hlo in [(256, 512, S // 4), (512, 1024, S // 8), (1024, 1024, S // 16), (128, 256, S // 2)]:
            hi, lo, w = rnd(B, chi, 2 * hlo, 2 * hlo), rnd(B, clo, hlo, hlo), rnd(clo, chi, 4, 4)
            dw = torch.empty_like(w)
            for split in (False, True):
                ops.set_split_gemm(split)
                pl = ops.wino_plan(B, chi, clo, hlo, hlo)
                assert pl is not None and pl.split == split, (pl, split)
                U, V, Mt = ops.wino_weights(w, pl), ops.wino_in(hi, pl), ops.wino_out_t(lo, pl)
                eb = 6 if split else 4
                tag = f"{'split' if split else 'fp32 '} hi {chi}@{2 * hlo} lo {clo}@{hlo} F{'42' if pl.variant else '22'}"
                for nm, fn, by in [("weights", lambda: ops.wino_weights(w, pl), 4 * w.numel() + (2 if split else 1) * eb * pl.nU),
                                   ("in", lambda: ops.wino_in(hi, pl), 4 * hi.numel() + eb * pl.nV),
                                   ("out_t", lambda: ops.wino_out_t(lo, pl), 4 * lo.numel() + eb * pl.nM)]:
                    ms = timeit(fn, R)
                    report(f"wino {nm:8s} {tag}", ms, 0, by)
                    acc(f"wino_{nm}_{'split' if split else 'fp32'}", ms)
                ops.profile_start()
                for _ in range(R):
                    ops.wino_down(U, V, pl); ops.wino_up(U, Mt, pl); ops.wino_wgrad(Mt, V, dw, pl)
                rec = ops.profile_stop()
                for nm in ("wfae_wino_gemm_down", "wfae_wino_gemm_up", "wfae_wino_gemm_wgrad"):
                    ms = rec[nm][1] / rec[nm][0]
                    report(f"{nm[5:]:19s} {tag}", ms, pl.gemm_flops, eb * (pl.nU + pl.nV) + 4 * pl.nM)
                    acc(f"{nm[5:]}_{'split' if split else 'fp32'}", ms)
                del U, V, Mt
            ops.set_split_gemm(True)
            del hi, lo, w, dw
    stages = [(256, S // 2), (512, S // 4), (1024, S // 8), (1024, S // 16), (1024, S // 8), (512, S // 4),
              (256, S // 2), (128, S)]
    if "conv1" in only:
        for c, h in stages:
            mid = c // 4
            for cin, cout, res in [(c, mid, False), (mid, c, True)]:
                x, w, dy = rnd(B, cin, h, h), rnd(cout, cin, 1, 1), rnd(B, cout, h, h)
                r = rnd(B, cout, h, h) if res else None
                dw = torch.empty_like(w)
                fl = 2 * B * h * h * cin * cout
                by = 4 * (B * h * h * (cin + cout) + cin * cout)
                for nm, fn, extra in [("fwd", lambda: ops.conv1x1_fwd(x, w, None, r), 4 * B * h * h * cout if res else 0),
                                      ("dgrad", lambda: ops.conv1x1_bwd_data(dy, w), 0),
                                      ("wgrad", lambda: ops.conv1x1_bwd_weight(dy, x, dw), 0)]:
                    ms = timeit(fn, R)
                    report(f"conv1x1 {nm:5s} {cin}->{cout} @{h}{' +res' if res and nm == 'fwd' else ''}", ms, fl, by + extra)
                    acc("conv1_" + nm, ms, 4)
                del x, w, dy, r, dw
    if "c1" in only:
        # csrc/c1r.hip (register-direct, exact three-plane bf16 operands) against gemm.hip's kernels, per Bottleneck stage it
        # serves: the four 1x1 products in the forms the step launches them (prologue / residual / BatchNorm sums)
        for c, h in [(256, S // 2), (128, S), (512, S // 4), (1024, S // 8), (1024, S // 16)]:
            mult = 4 if c == 128 or h == S // 16 else 8
            mid = c // 4
            n = B * h * h
            x, t2 = rnd(B, c, h, h), rnd(B, mid, h, h)
            w1, w3 = rnd(mid, c, 1, 1) * c ** -0.5, rnd(c, mid, 1, 1) * mid ** -0.5
            g, b_, rm, rv = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)
            st = ops.bn_stats_train(x, g, b_, rm, rv)
            g3, b3 = torch.ones(mid, device=dev), torch.zeros(mid, device=dev)
            st3 = ops.bn_stats_train(t2, g3, b3, torch.zeros(mid, device=dev), torch.ones(mid, device=dev))
            fl = 2 * n * c * mid
            by1, by3 = 4 * n * (c + mid), 4 * n * (c + mid) + 4 * n * c
            rows = [(f"fwd  {c}->{mid} bnact+stats", lambda: ops.conv1x1_fwd_bnact(x, st, w1, stats=True), fl, by1, "fwd1"),
                    (f"fwd  {c}->{mid} plain", lambda: ops.conv1x1_fwd(x, w1), fl, by1, "fwd1_plain"),
                    (f"dgrad {c}->{mid} (da3)", lambda: ops.conv1x1_bwd_data(x, w3), fl, by1, "dgrad3"),
                    (f"fwd  {mid}->{c} +res+stats", lambda: ops.conv1x1_fwd_stats(t2, w3, None, x), fl, by3, "fwd3"),
                    (f"fwd  {mid}->{c} bnact+res+stats", lambda: ops.conv1x1_fwd_bnact(t2, st3, w3, None, x, stats=True), fl, by3, "fwd3_bnact"),
                    (f"dgrad {mid}->{c} (da1)", lambda: ops.conv1x1_bwd_data(t2, w1), fl, by1, "dgrad1")]
            for tag, fn, f_, b2, key in rows:
                for on in (False, True):
                    ops.set_c1r(on)
                    ms = timeit(fn, R)
                    report(f"c1 @{h} {tag} {'c1r' if on else 'gemm.hip'}", ms, f_, b2)
                    acc(key + ("_c1r" if on else "_old"), ms, mult)
            ops.set_c1r(True)
            del x, t2, w1, w3
    if "bnseq" in only:
        # first BatchNorm of a Bottleneck, backward: data gradient of the C -> C/4 convolution + BatchNorm/GELU backward (+ residual
        # gradient), with the reduce pass separate or in the c1r epilogue (ops.set_c1r_bnred)
        from weatherforecastingtoolkit_amd import functional as Fn
        for c, h in [(256, S // 2), (128, S)]:
            mult = 4 if c == 128 else 8
            mid = c // 4
            n = B * h * h
            x, dt1, dy = rnd(B, c, h, h), rnd(B, mid, h, h), rnd(B, c, h, h)
            w1 = rnd(mid, c, 1, 1) * c ** -0.5
            g, b_, rm, rv = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)
            st = ops.bn_stats_train(x, g, b_, rm, rv)
            dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
            for rep in range(2):
                for on, dxf, nm in ((False, False, "reduce as a pass, dx as a pass"), (True, False, "reduce in the c1r epilogue, dx as a pass"),
                                    (True, True, "reduce and dx in c1r epilogues (dA recomputed)")):
                    ops.set_c1r_bnred(on)
                    ops.set_c1r_bndx(dxf)
                    ms = timeit(lambda: Fn._dgrad_bn(dt1, w1, None, x, g, st, dg, db, dy, True), R)
                    passes = 9 if not on else (7 if not dxf else 4)      # C-wide tensors moved
                    report(f"bnseq C={c} @{h} dgrad + bn1 bwd: {nm} rep{rep}", ms, 2 * n * c * mid * (2 if dxf else 1),
                           4 * n * (mid * (2 if dxf else 1) + passes * c))
                    acc("bnseq_" + ("separate" if not on else "fused" if not dxf else "recompute"), ms / 2, mult)
            ops.set_c1r_bnred(True)
            ops.set_c1r_bndx(True)
            if ops.c1r_bnred_supported(c, mid, h * h):     # the launches of the two c1r sequences on their own
                _, sr = ops.c1r_bnred(w1, dt1, x, st, store=False)
                ops.bn_act_bwd_from_rows(sr, c, dg, db)
                for nm, fn, by in (("c1r_bnred (stores dA)", lambda: ops.c1r_bnred(w1, dt1, x, st), 4 * n * (mid + 2 * c)),
                                   ("c1r_bnred (sums alone)", lambda: ops.c1r_bnred(w1, dt1, x, st, store=False), 4 * n * (mid + c)),
                                   ("c1r_bndx", lambda: ops.c1r_bndx(w1, dt1, x, g, st, dy, True), 4 * n * (mid + 3 * c)),
                                   ("bn_act_bwd_dx", lambda: ops.bn_act_bwd_dx(dy, x, g, st, dy, 1, True), 4 * n * 4 * c)):
                    report(f"bnseq C={c} @{h} {nm}", timeit(fn, R), 2 * n * c * mid, by)
            del x, dt1, dy
    if "c1b" in only:
        # bf16 storage ('medium'): csrc/c1b.hip against gemm.hip's bf16-storage kernels, per Bottleneck stage
        ops.set_float32_matmul_precision("medium")
        bf = torch.bfloat16
        for c, h in stages[:4] + stages[7:]:
            mult = 4 if c == 128 or h == S // 16 else 8
            mid = c // 4
            n = B * h * h
            x, t2 = rnd(B, c, h, h).to(bf), rnd(B, mid, h, h).to(bf)
            w1, w3 = rnd(mid, c, 1, 1) * c ** -0.5, rnd(c, mid, 1, 1) * mid ** -0.5
            g, b_, rm, rv = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)
            st = ops.bn_stats_train(x, g, b_, rm, rv)
            W1, W1t = ops.c1b_weights(w1)
            W3, W3t = ops.c1b_weights(w3)
            fl = 2 * n * c * mid
            by1, by3 = 2 * n * (c + mid), 2 * n * (c + mid) + 2 * n * c
            rows = [(f"fwd  {c}->{mid} bnact+stats c1b", lambda: ops.c1b_fwd(W1, x, st, None, True), fl, by1, "fwd1_c1b"),
                    (f"dgrad {c}->{mid} (da3) c1b", lambda: ops.c1b_fwd(W3t, x), fl, by1, "dgrad3_c1b"),
                    (f"fwd  {mid}->{c} +res+stats c1b", lambda: ops.c1b_fwd(W3, t2, None, x, True), fl, by3, "fwd3_c1b"),
                    (f"dgrad {mid}->{c} (da1) c1b", lambda: ops.c1b_fwd(W1t, t2), fl, by1, "dgrad1_c1b")]
            if ops.c1rb_supported(mid, c, h * h):
                rows += [(f"fwd  {c}->{mid} bnact+stats c1rb", lambda: ops.c1rb_fwd(w1, False, x, st, None, True), fl, by1, "fwd1_c1rb"),
                         (f"fwd  {c}->{mid} plain c1rb", lambda: ops.c1rb_fwd(w1, False, x), fl, by1, "fwd1_plain_c1rb"),
                         (f"dgrad {c}->{mid} (da3) c1rb", lambda: ops.c1rb_fwd(w3, True, x), fl, by1, "dgrad3_c1rb")]
            if ops.c1rb_supported(c, mid, h * h):
                rows += [(f"fwd  {mid}->{c} +res+stats c1rb", lambda: ops.c1rb_fwd(w3, False, t2, None, x, True), fl, by3, "fwd3_c1rb"),
                         (f"dgrad {mid}->{c} (da1) c1rb", lambda: ops.c1rb_fwd(w1, True, t2), fl, by1, "dgrad1_c1rb")]
            for tag, fn, f_, b2, key in rows:
                ms = timeit(fn, R)
                report(f"c1b @{h} {tag}", ms, f_, b2)
                acc(key, ms, mult)
            del x, t2, w1, w3, W1, W1t, W3, W3t
        ops.set_float32_matmul_precision(a.precision)
    if "c1w" in only:
        # weight gradients of the 1x1 convolutions: csrc/c1w.hip against gemm.hip, fp32 tensors and bf16 storage
        for prec, dt in (("highest", torch.float32), ("medium", torch.bfloat16)):
            ops.set_float32_matmul_precision(prec)
            tag = "fp32" if prec == "highest" else "bf16"
            es = 4 if prec == "highest" else 2
            for c, h in stages[:4] + stages[7:]:
                mult = 4 if c == 128 or h == S // 16 else 8
                mid = c // 4
                n = B * h * h
                x, dt1 = rnd(B, c, h, h).to(dt), rnd(B, mid, h, h).to(dt)
                g, b_, rm, rv = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)
                st = ops.bn_stats_train(x, g, b_, rm, rv)
                dw1, dw3 = torch.empty(mid, c, 1, 1, device=dev), torch.empty(c, mid, 1, 1, device=dev)
                fl, by = 2 * n * c * mid, es * n * (c + mid)
                for on in (False, True):
                    ops.set_c1w(on)
                    nm = "c1w" if on else "gemm.hip"
                    ms = timeit(lambda: ops.conv1x1_bwd_weight_bnact(dt1, x, st, dw1), R)
                    report(f"{tag} @{h} wgrad {c}->{mid} bnact {nm}", ms, fl, by)
                    acc(f"wgrad1_{tag}_{'c1w' if on else 'gemm'}", ms, mult)
                    ms = timeit(lambda: ops.conv1x1_bwd_weight(x, dt1, dw3), R)
                    report(f"{tag} @{h} wgrad {mid}->{c} {nm}", ms, fl, by)
                    acc(f"wgrad3_{tag}_{'c1w' if on else 'gemm'}", ms, mult)
                ops.set_c1w(True)
                del x, dt1
        ops.set_float32_matmul_precision(a.precision)
    if "fuse" in only:
        # BatchNorm-apply + GELU in the GEMM loaders vs the materialised form, first BatchNorm of a Bottleneck (C -> C/4)
        for c, h in stages:
            mid = c // 4
            x, w, dy = rnd(B, c, h, h), rnd(mid, c, 1, 1), rnd(B, mid, h, h)
            g, b_, rm, rv = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)
            st = ops.bn_stats_train(x, g, b_, rm, rv)
            dw = torch.empty_like(w)
            a = ops.bn_act_fwd(x, st, 1)
            fl, by = 2 * B * h * h * c * mid, 4 * (B * h * h * (c + mid) + c * mid)
            ms0 = timeit(lambda: ops.conv1x1_fwd(ops.bn_act_fwd(x, st, 1), w), R)
            ms1 = timeit(lambda: ops.conv1x1_fwd_bnact(x, st, w), R)
            report(f"bn+gelu ; conv1x1 fwd {c}->{mid} @{h}  (two kernels)", ms0, fl, by + 8 * x.numel())
            report(f"conv1x1_fwd_bnact     {c}->{mid} @{h}  (fused)", ms1, fl, by)
            ms2 = timeit(lambda: ops.conv1x1_bwd_weight(dy, a, dw), R)
            ms3 = timeit(lambda: ops.conv1x1_bwd_weight_bnact(dy, x, st, dw), R)
            report(f"conv1x1 wgrad {c}->{mid} @{h}  (saved a1)", ms2, fl, by)
            report(f"conv1x1_bwd_weight_bnact {c}->{mid} @{h}  (recompute)", ms3, fl, by)
            acc("fwd_two_kernels", ms0, 4); acc("fwd_fused", ms1, 4); acc("wgrad_saved", ms2, 4); acc("wgrad_recompute", ms3, 4)
            del x, w, dy, a, dw
    if "dconv" in only:
        for c, h in stages:
            mid = c // 4
            x, w, dy = rnd(B, mid, h, h), rnd(mid, mid // 8, 3, 3), rnd(B, mid, h, h)
            dw = torch.empty_like(w)
            fl = 2 * B * h * h * mid * (mid // 8) * 9
            by = 4 * 2 * B * h * h * mid
            for nm, fn in [("fwd", lambda: ops.gconv3x3_fwd(x, w, 8, False)),
                           ("dgrad", lambda: ops.gconv3x3_fwd(dy, w, 8, True)),
                           ("wgrad", lambda: ops.gconv3x3_bwd_weight(dy, x, dw, 8))]:
                ms = timeit(fn, R)
                report(f"gconv3x3 {nm:5s} {mid}ch g8 @{h}", ms, fl, by)
                acc("dconv_" + nm, ms, 4)
            del x, w, dy, dw
    if "g3b" in only:   # grouped 3x3 on bf16 tensors: dconv.hip kernels vs the implicit-GEMM kernel (csrc/g3b.hip)
        ops.set_float32_matmul_precision("medium")
        for c, h in stages:
            mid = c // 4
            x, w, dy = rnd(B, mid, h, h).bfloat16(), rnd(mid, mid // 8, 3, 3), rnd(B, mid, h, h).bfloat16()
            dw = torch.empty_like(w)
            fl = 2 * B * h * h * mid * (mid // 8) * 9
            by = 2 * 2 * B * h * h * mid
            for kern in ("dconv", "g3b"):
                ops.set_g3b(kern == "g3b")
                for nm, fn in [("fwd", lambda: ops.gconv3x3_fwd(x, w, 8, False)), ("dgrad", lambda: ops.gconv3x3_fwd(dy, w, 8, True))]:
                    ms = timeit(fn, R)
                    report(f"bf16 gconv3x3 {nm:5s} {mid}ch g8 @{h} {kern}", ms, fl, by)
                    acc(f"{kern}_{nm}", ms, 4)
                ms = timeit(lambda: ops.gconv3x3_bwd_weight(dy, x, dw, 8), R)
                report(f"bf16 gconv3x3 wgrad {mid}ch g8 @{h} {kern}", ms, fl, by)
                acc(f"{kern}_wgrad", ms, 4)
            del x, w, dy, dw
        ops.set_g3b(True)
        ops.set_float32_matmul_precision(a.precision)
    if "bn" in only:
        for c, h in stages:
            for ch in (c, c // 4):
                x, dy = rnd(B, ch, h, h), rnd(B, ch, h, h)
                g, b_, rm, rv = torch.ones(ch, device=dev), torch.zeros(ch, device=dev), torch.zeros(ch, device=dev), torch.ones(ch, device=dev)
                dg, db = torch.empty(ch, device=dev), torch.empty(ch, device=dev)
                n = x.numel()
                st = ops.bn_stats_train(x, g, b_, rm, rv)
                mult = 4 if ch == c else 8
                ms = timeit(lambda: ops.bn_stats_train(x, g, b_, rm, rv), R); report(f"bn_stats {ch}@{h}", ms, 0, 4 * n); acc("bn_stats", ms, mult)
                ms = timeit(lambda: ops.bn_act_fwd(x, st, 1), R); report(f"bn_act_fwd {ch}@{h}", ms, 0, 8 * n); acc("bn_fwd", ms, mult)
                ms = timeit(lambda: ops.bn_act_bwd(dy, x, g, st, dg, db, None, 1, True), R); report(f"bn_act_bwd {ch}@{h}", ms, 0, 20 * n); acc("bn_bwd", ms, mult)
                del x, dy
    print("---- per-step totals (ms), weighted by layer multiplicity ----")
    for k, v in tot.items():
        print(f"{k:14s} {v:9.2f}")


if __name__ == "__main__":
    main()
