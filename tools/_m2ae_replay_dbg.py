import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from gm3d_amd import engine_pretrain as E, point_m2ae as P
from bench import make_clouds
B = int(os.environ.get("B", "128"))
dp = os.environ.get("DP", "1") == "1"
P.FUSED_BLOCKS = os.environ.get("FB", "1") == "1"
torch.manual_seed(0)
model = P.PointM2AE().cuda().train()
if not dp:
    for m in model.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
ema = E.ModelEma(model, 0.999)
opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
args = SimpleNamespace(bf16=True, epochs=300)
pool = [make_clouds(B, 2048, 100 + i, "cuda") for i in range(4)]
for i in range(3):
    o = P.pretrain_step(model, ema, opt, pool[i].clone(), 100, args)
torch.cuda.synchronize()
print("eager loss", float(o["loss"]), "grad_norm", float(o["grad_norm"]))
static_in = pool[0].clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = P.pretrain_step(model, ema, opt, static_in, 100, args)
for i in range(4):
    static_in.copy_(pool[i % 4])
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}, "P finite", bool(torch.isfinite(opt.P).all()),
          "G finite", bool(torch.isfinite(opt.G).all()))
    if not torch.isfinite(opt.G).all():
        offs = list(opt._offs) + [opt.n]
        bad = [n for (n, p), o_, e in zip(opt._named, offs[:-1], offs[1:]) if not torch.isfinite(opt.G[o_:o_ + p.numel()]).all()]
        print("   non-finite grads:", bad[:12], len(bad))
        break
