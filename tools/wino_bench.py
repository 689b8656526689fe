"""Per-layer timing of the Winograd pieces (transforms vs GEMMs) at the bench shapes.
    python tools/wino_bench.py [--batch 32] [--size 384]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=384)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, S = a.batch, a.size
for chi, clo, hlo in [(256, 512, S // 4), (512, 1024, S // 8), (1024, 1024, S // 16), (128, 256, S // 2)]:
    hi = torch.rand(B, chi, 2 * hlo, 2 * hlo, device=dev) - 0.5
    lo = torch.rand(B, clo, hlo, hlo, device=dev) - 0.5
    w = torch.rand(clo, chi, 4, 4, device=dev) - 0.5
    dw = torch.empty_like(w)
    pl = ops.wino_plan(B, chi, clo, hlo, hlo)
    def run():
        U = ops.wino_weights(w, pl); V = ops.wino_in(hi, pl); Mt = ops.wino_out_t(lo, pl)
        ops.wino_down(U, V, pl); ops.wino_up(U, Mt, pl); ops.wino_wgrad(Mt, V, dw, pl)
    run(); torch.cuda.synchronize()
    ops.profile_start()
    for _ in range(3):
        run()
    prof = ops.profile_stop()
    print(f"--- hi {chi}@{2*hlo} lo {clo}@{hlo}  T={pl.T}  K4={4*chi}")
    for k, (calls, ms, fl, by) in sorted(prof.items()):
        print(f"   {k:24s} {ms / calls:8.3f} ms  {fl / ms / 1e9 if fl else 0:7.1f} TF  {by / ms / 1e6:8.0f} GB/s")
    del hi, lo, w, dw
