"""Per-stage GPU time of the pretrain step (HIP events), to direct optimisation.  python tools/stage_times.py [--fp32]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds

bf16 = "--fp32" not in sys.argv
B = 128
dev = torch.device("cuda")
torch.manual_seed(0)
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
opt = E.build_optimizer(model)
x0 = make_clouds(B, 1024, 1, dev)
amp = lambda: torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))

def step():
    marks.clear()
    mark("start")
    x = E.train_transforms(x0.clone()); mark("augment")
    t = ema.ema
    vis = torch.zeros(B, 64, dtype=torch.bool, device=dev)
    with amp():
        with torch.no_grad():
            group = t.group_divider(x); mark("fps+knn")
            nb, center, _ = group
            tok = t.encoder(nb); mark("T.embed")
            pos = t.pos_embed(center)
            xv = t.norm_p(t.blocks(tok, pos)); mark("T.enc12")
            lp = t.MAE_decoder_loss_pred(torch.cat([xv], 1), pos, 0)
            lpv = t._loss_pred_head(lp); mark("T.dec4+head")
            mask = t.generate_mask(lpv, 0.6, epoch=200, total_epoch=400).bool(); mark("mask")
        vis_ids, mask_ids = M.split_ids(mask, 25)
        tok = model.encoder(nb); mark("S.embed")
        pos = model.pos_embed(center)
        xv = model.norm_p(model.blocks(M.take(tok, vis_ids), M.take(pos, vis_ids))); mark("S.enc12")
        xf = torch.cat([xv, model.mask_token.expand(B, 39, -1).to(xv.dtype)], 1)
        pf = torch.cat([M.take(pos, vis_ids), M.take(pos, mask_ids)], 1)
        rec = model.MAE_decoder(xf, pf, 39)
        c = model.increase_dim_just_network_without_feature[0]
        pix = torch.nn.functional.linear(rec, c.weight.squeeze(-1), c.bias)
        lp = model._loss_pred_head(model.MAE_decoder_loss_pred(xf, pf, 39)); mark("S.dec2x4+heads")
        lo = model.forward_loss(pix[:, -39:], nb, mask)
        ll = model.forward_learning_loss(lp[:, -39:], mask, lo["matrix"].detach(), relative=True); mark("losses")
    opt.zero_grad(set_to_none=True)
    (lo["Chamfer_mean"] + ll).backward(); mark("backward")
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0, foreach=True); mark("clip")
    opt.step(); mark("adamw")
    ema.update(model); mark("ema")

for _ in range(3): step()
torch.cuda.synchronize()
acc = {}
N = 5
for _ in range(N):
    step(); torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = sum(acc.values())
for k, v in acc.items():
    print("%-16s %7.3f ms  %5.1f%%" % (k, v / N, 100 * v / tot))
print("total %.3f ms -> %.0f clouds/s" % (tot / N, B / (tot / N) * 1e3))
