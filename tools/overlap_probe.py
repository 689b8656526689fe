"""How well do an MFMA-bound GEMM and HBM-bound BN kernels overlap when issued on two streams?

    python tools/overlap_probe.py [wino|c1wgrad|c1fwd] [prio]      prio: BN stream gets high priority
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import ops
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "wino"
prio = len(sys.argv) > 2 and sys.argv[2] == "prio"
B = 32
hi, w = torch.rand(B, 512, 96, 96, device=dev), torch.rand(1024, 512, 4, 4, device=dev) * 0.01
x = torch.rand(B, 256, 192, 192, device=dev)
g, b_, rm, rv = (torch.ones(256, device=dev), torch.zeros(256, device=dev), torch.zeros(256, device=dev), torch.ones(256, device=dev))
st = ops.bn_stats_train(x, g, b_, rm, rv)
dy1, x1, dw1 = torch.rand(B, 128, 96, 96, device=dev), torch.rand(B, 512, 96, 96, device=dev), torch.empty(128, 512, 1, 1, device=dev)
w1 = torch.rand(128, 512, 1, 1, device=dev)
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream(priority=-1) if prio else torch.cuda.Stream()
def gemm(n=None):
    if which == "wino":
        for _ in range(n or 2): ops.conv4x4s2_down(hi, w)
    elif which == "c1wgrad":
        for _ in range(n or 30): ops.conv1x1_bwd_weight(dy1, x1, dw1)
    else:
        for _ in range(n or 30): ops.conv1x1_fwd(x1, w1)
def mem(n=26):
    for _ in range(n): ops.bn_act_fwd(x, st, 1)
def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    if fa:
        with torch.cuda.stream(s1): fa()
    if fb:
        with torch.cuda.stream(s2): fb()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
with torch.cuda.stream(s1): gemm(1)
with torch.cuda.stream(s2): mem(1)
for rep in range(3):
    a, b, c = timed(gemm, None), timed(None, mem), timed(gemm, mem)
    print(f"{which}{' prio' if prio else ''}: gemm alone {a:.2f} ms   bn alone {b:.2f} ms   both concurrently {c:.2f} ms   (sum {a+b:.2f}, max {max(a,b):.2f})", flush=True)
