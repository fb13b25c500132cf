"""Soak run of the captured step: N replays on rotating synthetic batches with host-to-device copies, allocations and a second
model's construction in between (the conditions under which a captured memset node went wrong); every 50 steps the
losses / gradient norm / parameters are checked for finiteness and printed.    python tools/soak.py [--steps 600]"""
import argparse, os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
E.enable_tuned_gemms()
torch.manual_seed(0)
dev = torch.device("cuda")
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
host = [make_clouds(a.batch, 1024, 50 + i, "cpu") for i in range(8)]
g = E.GraphedPretrainStep(model, ema, opt, args, host[0].to(dev), 200)
junk = []
for i in range(a.steps):
    E.adjust_learning_rate(opt, 200 + i / 1000.0, args)
    out = g(host[i % 8].to(dev))                       # a fresh host-to-device copy per step
    if i % 97 == 3:
        junk = [torch.nn.Linear(384, 1536).to(dev) for _ in range(8)] + [torch.randn(1 << 18).to(dev)]
    if i == 120:
        other = M.mae_vit_base_patch16_dec512d8b().to(dev)
    if i % 50 == 49 or i == a.steps - 1:
        vals = {k: float(out[k]) for k in ("loss", "loss_learn", "grad_norm")}
        ok = all(v == v and abs(v) != float("inf") for v in vals.values()) and bool(torch.isfinite(opt.P).all()) and bool(torch.isfinite(opt.E).all())
        print("step %4d %s %s" % (i + 1, vals, "ok" if ok else "NOT FINITE"), flush=True)
        if not ok:
            sys.exit(1)
print("soak ok")
