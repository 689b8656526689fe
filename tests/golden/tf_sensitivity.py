"""How far do the `_tf` model's transformer gradients move when only rounding changes?  Runs the reference module
(pipeline/models/ae_64x8x8_tf.PosAwareAE_TF, imported from /root/reference exactly as make_goldens.py g5 does:
train mode, dropout 0, B = 3) in fp32 and in fp64 on the G5 input and prints the max-relative difference of the
three gradient tensors tests/test_model_gpu.py::test_tf_variant_golden compares.  Batch-statistics BatchNorm at
B = 3 and attention across the batch amplify fp32 rounding; the test takes its tolerance from this output.

    python tests/golden/tf_sensitivity.py        (needs /root/reference; prints, writes nothing)
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("WFAE_REFERENCE", "/root/reference"))
from weatherforecastingtoolkit_amd import synth  # noqa: E402

torch.set_num_threads(8)
import pipeline.models.ae_64x8x8_tf as reftf  # noqa: E402

np_sd = synth.synth_state_dict(synth.ae_tf_state_dict_spec(128), seed=0)
x = torch.from_numpy(synth.uniform_frames(3, 128, seed=1234))
grads = {}
for dt in (torch.float32, torch.float64):
    net = reftf.PosAwareAE_TF()
    net.load_state_dict({k: (torch.from_numpy(np.asarray(v)) if v.ndim else torch.tensor(0)) for k, v in np_sd.items()},
                        strict=True)
    net = net.to(dt).train()
    for m in net.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0
    recon, z = net(x.to(dt))
    loss = F.l1_loss(recon, x.to(dt))
    loss.backward()
    grads[dt] = {"loss": loss.item(),
                 "g_tf0_inproj": net.tf.layers[0].self_attn.in_proj_weight.grad.double(),
                 "g_tf7_lin2": net.tf.layers[7].linear2.weight.grad.double()[:, :64],
                 "g_tf3_norm1": net.tf.layers[3].norm1.weight.grad.double()}
a, b = grads[torch.float32], grads[torch.float64]
print("loss fp32 / fp64:", a["loss"], b["loss"])
for k in ("g_tf0_inproj", "g_tf7_lin2", "g_tf3_norm1"):
    print(k, "max-rel fp32 vs fp64:", float((a[k] - b[k]).abs().max() / b[k].abs().max()))
