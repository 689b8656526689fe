"""How far do the per-parameter gradient norms of the AE move when only rounding changes?  Runs the oracle (the CPU
restatement that make_goldens.py asserts to be bit-identical to the reference modules) forward + L1 + backward in
fp32 and in fp64 on the G4 input (B = 1, 384 x 384) and prints the relative difference of every parameter's
gradient norm.  Batch-statistics BatchNorm over a single image makes the BatchNorm bias / weight gradients (sums
with heavy cancellation) the most sensitive entries; tests/test_model_gpu.py::test_full_model_golden takes its
gradient-norm tolerance from this output.

    python tests/golden/ae_gradnorm_sensitivity.py [batch] [size] [uniform|blobs]     (prints, writes nothing)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ae_oracle as orc  # noqa: E402
from weatherforecastingtoolkit_amd import synth  # noqa: E402

torch.set_num_threads(8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 384
np_sd = synth.synth_state_dict(synth.ae_state_dict_spec(S), seed=0)
if len(sys.argv) > 3 and sys.argv[3] == "blobs":      # the frames of the g3_*_blobs / g4_* fixtures (make_goldens.g_full)
    ev = synth.blob_events(1, S, B, seed=1234)
    x = torch.from_numpy((ev[0].transpose(2, 0, 1)[:, None].astype(np.float32)) * np.float32(1 / 255))
else:
    x = torch.from_numpy(synth.uniform_frames(B, S, seed=1234))
norms = {}
for dt in (torch.float32, torch.float64):
    sd = orc.to_torch_sd(np_sd)
    sd = {k: (v.detach().to(dt).requires_grad_(v.requires_grad) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    recon, z = orc.forward(x.to(dt), sd, True)
    loss = (recon - x.to(dt)).abs().mean()
    loss.backward()
    norms[dt] = {k: v.grad.double().norm().item() for k, v in sd.items() if getattr(v, "grad", None) is not None}
    print(dt, "loss", loss.item(), flush=True)
a, b = norms[torch.float32], norms[torch.float64]
rel = sorted(((abs(a[k] - b[k]) / (b[k] + 1e-30), k) for k in b), reverse=True)
print("largest gradient-norm deviations fp32 vs fp64:")
for r, k in rel[:24]:
    print(f"  {k:28s} {r:.3e}")
print("median", float(np.median([r for r, _ in rel])))
