import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import ops
dev = torch.device('cuda:0')
x = torch.rand(32, 128, 384, 384, device=dev) - 0.5
w = torch.rand(1, 128, 3, 3, device=dev) - 0.5
b = torch.rand(1, device=dev)
dy = torch.rand(32, 1, 384, 384, device=dev) - 0.5
dw = torch.empty_like(w)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fwd   %.3f ms" % t(lambda: ops.dconv_fwd(x, w, b, 3, 1, 1, 1)))
print("wgrad %.3f ms" % t(lambda: ops.dconv_bwd_weight(dy, x, dw, 3, 1, 1, 1)))
print("dgrad %.3f ms" % t(lambda: ops.dconv_bwd_data(dy, w, 128, 3, 1, 1)))
del x, dy
# first layer: Conv2d(1, 256, 4, stride 2, padding 1) on 384x384 frames
x1 = torch.rand(32, 1, 384, 384, device=dev)
w1 = torch.rand(256, 1, 4, 4, device=dev) - 0.5
dy1 = torch.rand(32, 256, 192, 192, device=dev) - 0.5
dw1 = torch.empty_like(w1)
print("first layer fwd   %.3f ms (ideal 0.19: 1.2 GB written)" % t(lambda: ops.dconv_fwd(x1, w1, None, 4, 2, 1, 1)))
print("first layer wgrad %.3f ms (ideal 0.19: 1.2 GB read)" % t(lambda: ops.dconv_bwd_weight(dy1, x1, dw1, 4, 2, 1, 1)))
