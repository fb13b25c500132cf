"""Stress: SegmentedDDPStep hipGraph replay vs the same three backward segments run eagerly, many times, world of one.
The flat gradient buffer after a replay must equal the eager one up to the atomics' summation order (<= 2e-3 of the segment max).
    python tools/seg_race.py [iters]      (environment switches of INTEGRATION.md section 6 apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from tests import clouds

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(os.environ.get("B", 4))
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
torch.manual_seed(0)
m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
for mod in m.modules():
    if isinstance(mod, M.DropPath):
        mod.drop_prob = 0.0
ema = E.ModelEma(m, 0.999)
opt = E.build_optimizer(m, lr=0.0, flat=True, model_ema=ema, segment_of=E.ddp_segment)      # lr 0: parameters never move
x0 = clouds.gaussian(B, 1024, 1).cuda()
seg = E.SegmentedDDPStep(m, ema, opt, args, x0, 200, warmup_iters=2, augment=False, inject_mask_noise=True)
worst = {0: 0.0, 1: 0.0, 2: 0.0}
bad = 0
for it in range(iters):
    x = clouds.gaussian(B, 1024, 100 + it).cuda()
    noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(it)).cuda()
    seg.static_noise.copy_(noise)
    seg._phase1(x.clone()); seg._phase2(); seg._phase3()
    seg._cut1 = seg._cut2 = seg._cut3 = None
    torch.cuda.synchronize()
    ref = opt.G.clone()
    opt.G.zero_()
    seg(x, noise)
    torch.cuda.synchronize()
    for sgm, (lo, hi) in opt.segment_ranges.items():
        e = float((opt.G[lo:hi] - ref[lo:hi]).abs().max() / ref[lo:hi].abs().max())
        worst[sgm] = max(worst[sgm], e)
        if e > 2e-3:
            bad += 1
            # which parameters?
            names = []
            for (n, p), o in zip(opt._named, opt._offs):
                if lo <= o < hi:
                    d = float((opt.G[o:o + p.numel()] - ref[o:o + p.numel()]).abs().max() / ref[lo:hi].abs().max())
                    if d > 2e-3:
                        names.append((n, round(d, 4)))
            print("iter %d segment %d err %.3e: %s" % (it, sgm, e, names[:6]))
print("worst per segment", worst, "bad", bad, "of", iters)
