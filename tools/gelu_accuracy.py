"""Accuracy of the GELU forms against fp64 (CPU only, no GPU needed): torch's fp32 CPU GELU (what the reference runs),
and an fp32 emulation (every operation rounded to fp32) of csrc/common.h::gelu_f — u * Phi(u) with the upper tail
1 - Phi(|u|) = t Q(t) exp(-u^2/2), t = 1/(1 + p|u|).  Also refits Q (Lawson-weighted least squares) to show where the
coefficients in common.h come from.

    python tools/gelu_accuracy.py
"""
import numpy as np
import torch
from scipy.special import erf

f32 = np.float32


def phi64(u):
    return 0.5 * (1 + erf(u.astype(np.float64) / np.sqrt(2)))


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def fit(n, p):
    ug = np.concatenate([np.linspace(0, 1, 4001), np.linspace(1, 6.2, 12001)])
    t = 1 / (1 + p * ug)
    target = (1 - phi64(ug)) * np.exp(ug * ug / 2)
    wgt = np.exp(-ug * ug / 2)
    a = np.vander(t, n + 1, increasing=True)[:, 1:]
    w = wgt.copy()
    for _ in range(40):
        c, *_ = np.linalg.lstsq(a * w[:, None], target * w, rcond=None)
        e = np.abs((a @ c - target) * wgt)
        w = w * (1 + 4 * e / e.max())
        w /= w.max()
    return c, e.max()


def cdf32(u, p, c):
    u = u.astype(f32)
    au = np.abs(u)
    one = np.ones_like(au)
    t = (f32(1) / fma(f32(p) * one, au, one)).astype(f32)
    e = np.exp2(((f32(-0.5) * u * u).astype(f32) * f32(1.4426950408889634)).astype(f32)).astype(f32)
    q = np.full_like(u, f32(c[-1]))
    for ck in c[-2::-1]:
        q = fma(q, t, np.full_like(u, f32(ck)))
    h = ((q * t).astype(f32) * e).astype(f32)
    return np.where(u >= 0, (f32(1) - h).astype(f32), h)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    u = np.concatenate([rng.uniform(-8, 8, 2_000_000), rng.normal(0, 1.5, 2_000_000), np.linspace(-9, 9, 400001)]).astype(f32)
    truth = u.astype(np.float64) * phi64(u)
    tg = torch.nn.functional.gelu(torch.from_numpy(u)).numpy().astype(np.float64)

    def rel(a):
        return np.abs(a - truth) / np.maximum(np.abs(truth), 1e-30)
    pos, neg = u > 0, (u < 0) & (u > -3)
    print(f"torch CPU fp32 F.gelu        : max abs err {np.abs(tg - truth).max():.3e}  rel err u>0 {rel(tg)[pos].max():.3e}  "
          f"rel err -3<u<0 {rel(tg)[neg].max():.3e}")
    for n, p in [(5, 0.2316419), (6, 0.275), (7, 0.24)]:
        c, me = fit(n, p)
        g = (u * cdf32(u, p, c)).astype(f32).astype(np.float64)
        print(f"tail form degree {n} p={p:<9}: fit err {me:.1e}  max abs err {np.abs(g - truth).max():.3e}  rel err u>0 "
              f"{rel(g)[pos].max():.3e}  rel err -3<u<0 {rel(g)[neg].max():.3e}  max |this - torch| {np.abs(g - tg).max():.3e}")
        if n == 6:
            print("   coefficients t^1..t^6 (csrc/common.h):", ", ".join(f"{x:.16g}" for x in c))
