import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weatherforecastingtoolkit_amd import synth, functional as Fn
from weatherforecastingtoolkit_amd.pipeline.models.ae_64x8x8_lin import PosAwareAE_TF
from weatherforecastingtoolkit_amd.optim import FusedAdamW, CosineWarmupLR
from oracle import ae_oracle as orc
dev = torch.device("cuda:0")
np_sd = synth.synth_state_dict(synth.ae_state_dict_spec(128), seed=0)
x = torch.from_numpy(synth.uniform_frames(2, 128, seed=1234))
net = PosAwareAE_TF().to(dev)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in np_sd.items()}, strict=True)
net.train()
opt = FusedAdamW(net.parameters(), lr=5e-5, weight_decay=1e-4)
sch = CosineWarmupLR(opt, 5e-6, 5e-7, 5e-5, 40, 4.0)
sd = orc.to_torch_sd(np_sd)
oopt = orc.make_optimizer([p for _, p in orc.trainable(sd)], lr=5e-5, weight_decay=1e-4)
osch = orc.make_scheduler(oopt, 5e-6, 5e-7, 5e-5, 40, 4.0)
xd = x.to(dev)
for s in range(3):
    opt.zero_grad(set_to_none=True)
    recon, z = net(xd); loss = Fn.l1_loss(recon, xd); loss.backward(); opt.step(); sch.step()
    orc.train_step(x, sd, oopt, osch)
    msd = net.state_dict()
    worst = []
    for k, v in sd.items():
        a, b = msd[k].detach().double().cpu(), v.detach().double()
        if a.ndim == 0:
            if int(a) != int(b): print("NBT mismatch", k, int(a), int(b))
            continue
        worst.append((float((a - b).abs().max() / (b.abs().max() + 1e-30)), k))
    worst.sort(reverse=True)
    print(f"step {s}: worst state entries:", worst[:6])
net.eval()
with torch.no_grad():
    er, ez = net(xd)
    orr, oz = orc.forward(x, sd, training=False)
print("eval z relerr", float((ez.cpu() - oz).abs().max() / oz.abs().max()), "recon", float((er.cpu() - orr).abs().max() / orr.abs().max()))
# eval forward with the ORACLE's state loaded into my net
net.load_state_dict({k: v.detach() for k, v in sd.items()}, strict=True)
with torch.no_grad():
    er2, ez2 = net(xd)
print("eval (oracle state) z relerr", float((ez2.cpu() - oz).abs().max() / oz.abs().max()), "recon", float((er2.cpu() - orr).abs().max() / orr.abs().max()))
# layerwise eval of encoder
with torch.no_grad():
    h = xd; ho = x
    for i in range(4):
        h = net.enc[i](h); ho = orc.enc_block(ho, sd, f"enc.{i}", False)
        print("enc", i, float((h.cpu() - ho).abs().max() / ho.abs().max()))
