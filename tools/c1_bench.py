import sys, torch
sys.path.insert(0, '/root/repo')
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
