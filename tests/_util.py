import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def relerr(a, b):
    """max |a-b| / (max |b| + tiny), both numpy or torch (moved to cpu)."""
    import torch
    if isinstance(a, torch.Tensor):
        a = a.detach().double().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().double().cpu().numpy()
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def l1_backward_on_reference_branch(recon, x, g, replicas=1, max_flips=16):
    """Backward pass of mean |recon - x| on the REFERENCE's side of the L1 kink.

    d loss / d recon = sign(recon - x) / N jumps at recon == x, and a pixel whose |recon - x| is below the forward
    tolerance (1e-4 relative) may legitimately land on the other side: at B = 4, 384x384 one pixel has |recon - x| =
    1.4e-5 in the reference, at B = 1 one has 4.3e-6.  A single flipped sign is a rank-one change of the loss gradient
    that the backward pass spreads over every encoder parameter (tools/debug_bwd_chain.py, tools/debug_split_kink.py:
    gradient norms move by up to 4e-3 although every kernel matches to 5e-5 on the same branch).  So (1) the sign
    patterns may differ from the reference's (stored in the fixture) only on near-ties, (2) the gradients are compared on
    the reference's branch, (3) the product's own backward kernel is checked against the same pattern.  `replicas`: the batch is the fixture's batch repeated that many times."""
    import torch
    n_one = recon.numel() // replicas
    gt = torch.from_numpy(np.unpackbits(g["l1_gt_bits"])[:n_one].astype(np.float32)).to(recon.device)
    lt = torch.from_numpy(np.unpackbits(g["l1_lt_bits"])[:n_one].astype(np.float32)).to(recon.device)
    ref_sign = (gt - lt).view((recon.shape[0] // replicas,) + tuple(recon.shape[1:])).repeat(replicas, 1, 1, 1)
    d = recon.detach() - x
    flips = torch.sign(d) != ref_sign
    assert int(flips.sum()) <= max_flips * replicas, int(flips.sum())
    assert bool((d.abs()[flips] <= 1e-4 * x[flips].abs().clamp(min=1e-3)).all()), "a sign differs away from a tie"
    # (3) the PRODUCT's own L1 backward kernel (wfae_l1_bwd, what loss.backward() launches) at this size: its
    # d loss / d recon must equal the reference's sign / N BIT FOR BIT on every pixel that is not one of the (<= 16 per
    # image) near-ties above, and +-1/N with the product's own sign on those
    from weatherforecastingtoolkit_amd import ops
    own = ops.l1_bwd(recon.detach().contiguous(), x.contiguous(), torch.ones((), device=recon.device), 1.0)
    want = ref_sign / recon.numel()
    assert torch.equal(own[~flips], want[~flips]), "wfae_l1_bwd differs from sign(recon - x) / N away from the near-ties"
    assert torch.equal(own[flips], (torch.sign(d) / recon.numel())[flips])
    recon.backward(want)
