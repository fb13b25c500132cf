"""diagnostic: per-parameter gradient error of gm3d_amd.point_m2ae (GPU, fp32) and of the fp32 CPU oracle against the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, point_m2ae as P
from oracle import hier_ref as HR, model_ref as R, ops as O
from tests import clouds
O.build()
if os.environ.get("NOCUDNN"): torch.backends.cudnn.enabled = False
B, total, epoch = 2, 300, 0
pts = clouds.gaussian(B, 2048, seed=21)
noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(5))
model = P.PointM2AE()
for mod in model.modules():
    if hasattr(mod, "drop_prob"):
        mod.drop_prob = 0.0
R.det_fill_(model, seed=3)
sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
teacher_sd = {k: v.clone() for k, v in R.det_fill_(P.PointM2AE(), seed=4).state_dict().items()}
model = model.cuda().train()
ema = E.ModelEma(model, 0.999)
ema.ema.load_state_dict(teacher_sd)
ptaps = {}
for i in range(3):
    model.token_embed[i].register_forward_hook(lambda m, inp, o, i=i: (o.retain_grad(), ptaps.__setitem__("tok%d" % i, o))[1])
out = P.pretrain_forward(model, ema.ema, pts.cuda(), epoch, total, mask_noise=noise.cuda())
out["loss"].backward()
ref = HR.m2ae_pretrain_forward(sd, teacher_sd, pts, epoch, total, noise)
ref["loss"].backward()
sd64 = {k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point else v.detach()) for k, v in sd.items()}
t64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in teacher_sd.items()}
taps64 = {}
ref64 = HR.m2ae_pretrain_forward(sd64, t64, pts, epoch, total, noise, mask=ref["mask"], taps=taps64)
ref64["loss"].backward()
for k in sorted(taps64):
    a, b = ptaps[k].grad.cpu().double(), taps64[k].grad
    print(k, "value err %.2e" % float((ptaps[k].detach().cpu().double() - taps64[k].detach()).abs().max() / taps64[k].detach().abs().max()),
          "grad err %.2e" % float((a - b).abs().max() / b.abs().max()), "grad max", float(b.abs().max()))
    if k == "tok1":
        masks = HR.propagate_visibility(ref["mask"], HR.hierarchical_group(pts)[2])
        err = (a - b).abs().amax(dim=-1)                       # (B,256)
        top = torch.topk(err.flatten(), 10)
        for v, idx in zip(top.values.tolist(), top.indices.tolist()):
            bb, tt = divmod(idx, err.shape[1])
            print("   token (%d,%d) err %.2e  masked=%s  |g64|max %.2e |gprod|max %.2e" % (bb, tt, v, bool(masks[1][bb, tt]),
                  float(b[bb, tt].abs().max()), float(a[bb, tt].abs().max())))
        print("   masked tokens: max err %.2e; visible tokens: max err %.2e" % (float(err[masks[1]].max()), float(err[~masks[1]].max())))
m0 = P.back_project(out["mask"], [x.cuda() for x in ref64["rec"].new_zeros(1).long().new_zeros(1).unsqueeze(0)] ) if False else None
gscale = max(float(t.grad.abs().max()) for t in sd64.values() if torch.is_tensor(t) and t.grad is not None)
rows = []
for name, p in model.named_parameters():
    g64 = sd64[name].grad
    if p.grad is None or g64 is None:
        continue
    scale = max(float(g64.abs().max()), 1e-4 * gscale)
    rows.append((float((p.grad.cpu().double() - g64).abs().max()) / scale, float((sd[name].grad.double() - g64).abs().max()) / scale, name))
for e, c, n in rows:
    if e > 2e-5:
        print("%-55s prod %.2e  cpu32 %.2e" % (n, e, c))
