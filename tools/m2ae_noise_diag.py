"""Is the residual between the Point-M2AE product gradient (GPU fp32) and the decision-injected fp64 oracle rounding noise of the
GPU's fp32 reductions, or a branch difference?  The same clouds in the opposite batch order give the same gradient in exact
arithmetic; every BatchNorm / column reduction then sums its rows in another order.  Prints, per parameter with the largest
difference, |grad(order A) - grad(order B)| relative to the tensor's max, and whether the two runs took the same decisions.
    python tools/m2ae_noise_diag.py SEED [SEED ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gm3d_amd import engine_pretrain as E, point_m2ae as P  # noqa: E402
from oracle import model_ref as R  # noqa: E402
from tests import clouds  # noqa: E402


def run(seed, flip):
    epoch, total = (0 if seed % 2 == 0 else 200), 300
    pts = clouds.gaussian(2, 2048, seed=seed)
    noise = torch.rand(2, 64, generator=torch.Generator().manual_seed(5))
    if flip:
        pts, noise = pts.flip(0).contiguous(), noise.flip(0).contiguous()
    model = P.PointM2AE()
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    R.det_fill_(model, seed=3)
    teacher_sd = {k: v.clone() for k, v in R.det_fill_(P.PointM2AE(), seed=4).state_dict().items()}
    model = model.cuda().train()
    ema = E.ModelEma(model, 0.999)
    ema.ema.load_state_dict(teacher_sd)
    P.POOL_TAPS, P.ACT_TAPS = [], []
    out = P.pretrain_forward(model, ema.ema, pts.cuda(), epoch, total, mask_noise=noise.cuda())
    taps = [t.clone() for t in P.POOL_TAPS + P.ACT_TAPS]
    P.POOL_TAPS = P.ACT_TAPS = None
    out["loss"].backward()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, taps, out["mask"].clone()


for seed in [int(a) for a in sys.argv[1:]] or [25, 28]:
    ga, ta, ma = run(seed, False)
    gb, tb, mb = run(seed, True)
    same_mask = torch.equal(ma, mb.flip(0))
    ndiff = 0
    for a, b in zip(ta, tb):
        # taps are (B*G, C) / (B*G*k, C) row-major over the batch: undo the flip by halves
        h = b.shape[0] // 2
        b2 = torch.cat([b[h:], b[:h]])
        ndiff += int((a != b2).sum())
    gs = max(float(v.abs().max()) for v in ga.values())
    rows = sorted(((float((ga[n] - gb[n]).abs().max()) / max(float(ga[n].abs().max()), 1e-4 * gs), n) for n in ga), reverse=True)
    print("seed %d: same mask %s, %d differing decisions between the two batch orders; largest gradient differences:" % (seed, same_mask, ndiff))
    for r, n in rows[:10]:
        print("     %.3e  %s" % (r, n))
