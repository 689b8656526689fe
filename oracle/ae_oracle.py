"""ORACLE — test infrastructure only, never the product path.

CPU fp32 restatement of the reference's conv-autoencoder training hot path,
written functionally over a flat ``state_dict`` (no nn.Module tree), with the
torch CPU built-ins the reference itself executes (F.conv2d, F.batch_norm,
F.gelu, F.linear, F.l1_loss, torch.optim.AdamW).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (`weatherforecastingtoolkit_amd`) never does.

Pinned: ``tests/golden/*.npz`` were produced in the build container by
importing the real reference modules from /root/reference
(``tests/golden/make_goldens.py``) and ``tests/test_oracle_golden.py`` checks
this restatement against them.  SSIM/PSNR follow the published defaults of
pytorch_msssim / torchmetrics, which are not installed and not vendored in the
reference: that part is "parity unpinned" (SURVEY.md §8c).

Reference lines followed (relative to /root/reference):
  Bottleneck            pipeline/models/ae_64x8x8_lin.py:7-22
  EncBlock / DecBlock   pipeline/models/ae_64x8x8_lin.py:27-47
  PosAwareAE_TF         pipeline/models/ae_64x8x8_lin.py:52-106
  Loss.forward (live)   experiments/ae_v2/train.py:54-74
  adamw_optimizer       pipeline/helpers.py:63-74
  cosine_warmup         pipeline/helpers.py:76-107
  metrics.ssim / psnr   pipeline/metrics.py:71-93
  NLayerDiscriminator   pipeline/models/autoencoderkl/losses/model.py:100-150
  hinge_d_loss          pipeline/models/autoencoderkl/losses/contperceptual.py:19-23
  adaptive weight, G/D  experiments/ae_v2_2/train.py:54-95 (Loss), :126-168 (training_step)
  linear forecaster     experiments/v1_experiments/pretrained_ae_linear_sevir/train.py:67,73-83
                        ("parity unpinned": that script cannot be imported here — wandb / lightning / diffusers —
                        and the reference holds no fixture for it; the restatement is the same torch calls)
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ---------------------------------------------------------------- blocks ---
def _bn(x, sd, p, training):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1) — ae_64x8x8_lin.py:14,16,18,32,43.
    In training mode updates running stats in ``sd`` in place like the module."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training:
        sd[p + ".num_batches_tracked"] += 1
    return y


def bottleneck(x, sd, p, training, groups=8):
    """x + f(x), f = BN,GELU,1x1,BN,GELU,g3x3,BN,GELU,1x1 — ae_64x8x8_lin.py:7-22."""
    c = x.shape[1]
    mid = c // 4
    g = min(groups, mid)
    h = F.gelu(_bn(x, sd, p + ".f.0", training))
    h = F.conv2d(h, sd[p + ".f.2.weight"])
    h = F.gelu(_bn(h, sd, p + ".f.3", training))
    h = F.conv2d(h, sd[p + ".f.5.weight"], padding=1, groups=g)
    h = F.gelu(_bn(h, sd, p + ".f.6", training))
    h = F.conv2d(h, sd[p + ".f.8.weight"])
    return x + h


def _num_res(sd, p):
    j = 0
    while f"{p}.res.{j}.f.2.weight" in sd:
        j += 1
    return j


def enc_block(x, sd, p, training, groups=8):
    """Conv2d(4,s2,p1,no bias) -> BN -> GELU -> bottlenecks — ae_64x8x8_lin.py:27-36."""
    h = F.conv2d(x, sd[p + ".down.0.weight"], stride=2, padding=1)
    h = F.gelu(_bn(h, sd, p + ".down.1", training))
    for j in range(_num_res(sd, p)):
        h = bottleneck(h, sd, f"{p}.res.{j}", training, groups)
    return h


def dec_block(x, sd, p, training, groups=8):
    """ConvTranspose2d(4,s2,p1,no bias) -> BN -> GELU -> bottlenecks — :38-47."""
    h = F.conv_transpose2d(x, sd[p + ".up.0.weight"], stride=2, padding=1)
    h = F.gelu(_bn(h, sd, p + ".up.1", training))
    for j in range(_num_res(sd, p)):
        h = bottleneck(h, sd, f"{p}.res.{j}", training, groups)
    return h


def _count(sd, fmt):
    i = 0
    while fmt.format(i) in sd:
        i += 1
    return i


def encode(x, sd, training, groups=8):
    """enc(x) + pos_emb -> flatten -> to_latent — ae_64x8x8_lin.py:88-94."""
    n_enc = _count(sd, "enc.{}.down.0.weight")
    h = x
    for i in range(n_enc):
        h = enc_block(h, sd, f"enc.{i}", training, groups)
    h = F.conv2d(h, sd[f"enc.{n_enc}.weight"], sd[f"enc.{n_enc}.bias"])
    h = h + sd["pos_emb"]
    return F.linear(h.flatten(1), sd["to_latent.weight"], sd["to_latent.bias"])


def decode(z, sd, training, groups=8):
    """from_latent -> view(B,C,hw,hw) -> dec -> sigmoid — ae_64x8x8_lin.py:96-102.
    hw comes from pos_emb (8 in the reference; 24 for the 384 extension)."""
    pe = sd["pos_emb"]
    h = F.linear(z, sd["from_latent.weight"], sd["from_latent.bias"])
    h = h.view(z.shape[0], pe.shape[1], pe.shape[2], pe.shape[3])
    h = F.conv2d(h, sd["dec.0.weight"], sd["dec.0.bias"])
    k = 1
    while f"dec.{k}.up.0.weight" in sd:
        h = dec_block(h, sd, f"dec.{k}", training, groups)
        k += 1
    h = F.conv2d(h, sd[f"dec.{k}.weight"], sd[f"dec.{k}.bias"], padding=1)
    return torch.sigmoid(h)


def forward(x, sd, training=True, groups=8):
    """returns (recon, z) — ae_64x8x8_lin.py:104-106."""
    z = encode(x, sd, training, groups)
    return decode(z, sd, training, groups), z


# ------------------------------------------------------------------ loss ---
def _gauss_1d(size=11, sigma=1.5, dtype=torch.float32):
    c = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(x, y, data_range=1.0, size_average=True):
    """pytorch_msssim.ssim defaults (call site experiments/ae_v2/train.py:62):
    11-tap Gaussian sigma 1.5, valid window, K=(0.01,0.03); mean over the map
    per (n,c) then mean.  UNPINNED third-party arithmetic."""
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    ch = x.shape[1]
    g = _gauss_1d(dtype=x.dtype).to(x.device)
    wv = g.view(1, 1, -1, 1).repeat(ch, 1, 1, 1)
    wh = g.view(1, 1, 1, -1).repeat(ch, 1, 1, 1)

    def filt(t):
        return F.conv2d(F.conv2d(t, wv, groups=ch), wh, groups=ch)

    mu1, mu2 = filt(x), filt(y)
    s11 = filt(x * x) - mu1 * mu1
    s22 = filt(y * y) - mu2 * mu2
    s12 = filt(x * y) - mu1 * mu2
    cs = (2 * s12 + c2) / (s11 + s22 + c2)
    m = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs
    per = m.flatten(2).mean(-1)
    return per.mean() if size_average else per.mean(1)


def psnr(pred, target):
    """pipeline/metrics.py:77-84: torchmetrics PeakSignalNoiseRatio() with
    data_range=None per single-sample call => range = max(target)-min(target)
    of that sample; mean over samples.  UNPINNED (torchmetrics absent)."""
    tot = 0.0
    for i in range(pred.shape[0]):
        p, g = pred[i], target[i]
        mse = torch.mean((p - g) ** 2)
        r = g.max() - g.min()
        tot += float(10.0 * torch.log10(r * r / mse))
    return tot / pred.shape[0]


def loss_fn(recon, x, recon_weight=1.0, perceptual_weight=0.0):
    """Live branch of Loss.forward — experiments/ae_v2/train.py:54-74."""
    rec = recon_weight * F.l1_loss(recon, x, reduction="mean")
    if perceptual_weight > 0:
        rec = rec + perceptual_weight * (1 - ssim(x.repeat(1, 3, 1, 1), recon.repeat(1, 3, 1, 1), 1.0))
    return rec


# ------------------------------------------------------------- optimiser ---
def lr_at(step, start_lr, peak_lr, final_lr, total_steps, warmup_steps):
    """LR after `step` scheduler.step() calls of the SequentialLR built by
    cosine_warmup_scheduler (pipeline/helpers.py:76-107), in closed form.
    warmup_steps may be non-integer (train.py:258): the milestone is then never
    hit exactly, torch enters CosineAnnealingLR through its chained form and the
    ratios telescope to (1+cos(pi t/T))/(1+cos(pi/T)) from the last linear LR.
    Pinned against the real torch classes by tests/golden/g8_sched.npz."""
    sf = start_lr / peak_lr
    lin = lambda k: peak_lr * (sf + (1.0 - sf) * min(k, warmup_steps) / warmup_steps)
    if step < warmup_steps:
        return lin(step)
    tmax = total_steps - warmup_steps
    if float(warmup_steps).is_integer():
        t = step - int(warmup_steps)
        return final_lr + (peak_lr - final_lr) * (1 + math.cos(math.pi * t / tmax)) / 2
    first = math.floor(warmup_steps) + 1
    t = step - first
    return final_lr + (lin(first - 1) - final_lr) * (1 + math.cos(math.pi * t / tmax)) / (1 + math.cos(math.pi / tmax))


def make_optimizer(params, lr=5e-5, weight_decay=1e-4, beta1=0.9, beta2=0.999):
    """adamw_optimizer — pipeline/helpers.py:63-74."""
    return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=(beta1, beta2))


def make_scheduler(opt, start_lr, final_lr, peak_lr, total_steps, warmup_steps):
    """cosine_warmup_scheduler — pipeline/helpers.py:76-107 (same torch classes)."""
    for g in opt.param_groups:
        g["lr"] = peak_lr
    w = torch.optim.lr_scheduler.LinearLR(opt, start_factor=start_lr / peak_lr, end_factor=1.0, total_iters=warmup_steps)
    c = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=total_steps - warmup_steps, eta_min=final_lr)
    return torch.optim.lr_scheduler.SequentialLR(opt, schedulers=[w, c], milestones=[warmup_steps])


# ------------------------------------------------------------ train step ---
def to_torch_sd(np_sd, requires_grad=True):
    sd = OrderedDict()
    for k, v in np_sd.items():
        t = torch.from_numpy(v.copy()) if v.ndim else torch.tensor(int(v), dtype=torch.int64)
        if requires_grad and t.dtype.is_floating_point and "running_" not in k:
            t.requires_grad_(True)
        sd[k] = t
    return sd


def trainable(sd):
    return [(k, v) for k, v in sd.items() if v.requires_grad]


def train_step(x, sd, opt=None, sched=None, recon_weight=1.0, perceptual_weight=0.0):
    """fwd -> loss -> bwd -> AdamW -> LR schedule, the order of
    Model.training_step + Lightning (experiments/ae_v2/train.py:209-223,254-261)."""
    for _, p in trainable(sd):
        p.grad = None
    recon, z = forward(x, sd, training=True)
    loss = loss_fn(recon, x, recon_weight, perceptual_weight)
    loss.backward()
    if opt is not None:
        opt.step()
        if sched is not None:
            sched.step()
    return recon.detach(), z.detach(), float(loss.detach())


# ------------------------------------------------------------- AE + GAN ---
def disc_forward(x, sd, training=True, p="main"):
    """NLayerDiscriminator.forward — autoencoderkl/losses/model.py:123-150: Conv(4,s2,p1,bias)+LReLU,
    [Conv(4,s2,p1)+BN+LReLU]*(n-1), Conv(4,s1,p1)+BN+LReLU, Conv(1x1, padding=1, bias)."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith(p + ".") and k.endswith(".weight") and sd[k].ndim == 4})
    h = F.leaky_relu(F.conv2d(x, sd[f"{p}.{idx[0]}.weight"], sd[f"{p}.{idx[0]}.bias"], stride=2, padding=1), 0.2)
    for j, i in enumerate(idx[1:-1]):
        stride = 1 if j == len(idx) - 3 else 2
        h = F.conv2d(h, sd[f"{p}.{i}.weight"], None, stride=stride, padding=1)
        h = F.leaky_relu(_bn(h, sd, f"{p}.{i + 1}", training), 0.2)
    return F.conv2d(h, sd[f"{p}.{idx[-1]}.weight"], sd[f"{p}.{idx[-1]}.bias"], stride=1, padding=1)


def hinge_d_loss(logits_real, logits_fake):
    """contperceptual.py:19-23"""
    return 0.5 * (torch.mean(F.relu(1.0 - logits_real)) + torch.mean(F.relu(1.0 + logits_fake)))


def adaptive_weight(rec_loss, g_loss, last_layer, disc_weight=1.0):
    """Loss.calculate_adaptive_weight — experiments/ae_v2_2/train.py:46-52"""
    rec_grad = torch.autograd.grad(rec_loss, last_layer, retain_graph=True)[0]
    disc_grad = torch.autograd.grad(g_loss, last_layer, retain_graph=True)[0]
    d_weight = torch.norm(rec_grad) / (torch.norm(disc_grad) + 1e-4)
    return torch.clamp(disc_weight * d_weight, 0.0, 1e4).detach()


def gan_train_step(x, sd, dsd, g_opt, d_opt, g_sched=None, d_sched=None, gan_on=True, disc_weight=1.0,
                   recon_weight=1.0, clip=1.0):
    """one manual-optimisation step of experiments/ae_v2_2/train.py:126-159: forward once; generator
    loss (L1 + d_weight * -mean(D(x_hat))) with the discriminator frozen (Lightning toggle_optimizer),
    backward, clip-by-norm, AdamW + scheduler; then hinge discriminator loss on (x, x_hat.detach()),
    backward, clip, AdamW + scheduler.  gan_on = (global_step >= disc_start)."""
    ae_params = [q for _, q in trainable(sd)]
    d_params = [q for _, q in trainable(dsd)]
    log = {}
    recon, z = forward(x, sd, training=True)
    # ---- generator
    for q in d_params:
        q.requires_grad_(False)
    rec_loss = recon_weight * F.l1_loss(recon, x, reduction="mean")
    if gan_on:
        g_loss = -torch.mean(disc_forward(recon, dsd, True))
        d_weight = adaptive_weight(rec_loss, g_loss, sd[_last_layer_key(sd)], disc_weight)
        loss = rec_loss + d_weight * g_loss
        log.update(g_loss=float(g_loss.detach()), d_weight=float(d_weight))
    else:
        loss = rec_loss
    log.update(rec_loss=float(rec_loss.detach()), total_loss=float(loss.detach()))
    loss.backward()
    log["g_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(ae_params, clip))
    g_opt.step()
    if g_sched is not None:
        g_sched.step()
    for q in ae_params:
        q.grad = None
    for q in d_params:
        q.requires_grad_(True)
    # ---- discriminator
    if gan_on:
        logits_real = disc_forward(x.detach(), dsd, True)
        logits_fake = disc_forward(recon.detach(), dsd, True)
        d_loss = hinge_d_loss(logits_real, logits_fake)
        d_loss.backward()
        log.update(disc_loss=float(d_loss.detach()), logits_real=float(logits_real.detach().mean()),
                   logits_fake=float(logits_fake.detach().mean()))
        log["d_grad_norm"] = float(torch.nn.utils.clip_grad_norm_(d_params, clip))
        d_opt.step()
        if d_sched is not None:
            d_sched.step()
        for q in d_params:
            q.grad = None
    return recon.detach(), log


def _last_layer_key(sd):
    """Model.get_last_layer — experiments/ae_v2_2/train.py:123-124: autoencoder.dec[-1].weight"""
    n = max(int(k.split(".")[1]) for k in sd if k.startswith("dec."))
    return f"dec.{n}.weight"


# ------------------------------------------------ Path-B linear forecaster ---
def linear_forecast_loss(v, weight, bias, input_frames):
    """Model.training_step of v1_experiments/pretrained_ae_linear_sevir/train.py:73-82 on latents v (B,T,C,h,w):
    difference against the last input frame, per-latent-pixel nn.Linear(Tin*C, Tout*C), F.mse_loss.
    (The reference hard-codes 13 / 12 frames in the reshapes; here they follow input_frames / T.)"""
    b, t, c, h, w = v.shape
    tout = t - input_frames
    inp, tgt = v[:, :input_frames], v[:, input_frames:]
    inp_t = inp[:, -1].unsqueeze(1)
    inp = inp - inp_t
    tgt = tgt - inp_t
    x = inp.permute(0, 3, 4, 1, 2).reshape(b, h, w, input_frames * c)
    pred = F.linear(x, weight, bias).permute(0, 3, 1, 2).reshape(b, tout, c, h, w)
    return F.mse_loss(pred, tgt), pred + inp_t
