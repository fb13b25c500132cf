"""Full Point-M2AE step (with clip + AdamW + EMA) eager vs captured/replayed: which outputs go wrong on replay?"""
import argparse, os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, point_m2ae as P
from bench import make_clouds

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--twin-idle", action="store_true", help="create the twin but never step it")
ap.add_argument("--junk", type=int, nargs=2, default=None, help="after the capture allocate NaN-filled tensors of lo..hi BYTES (x2 steps, 64 each)")
ap.add_argument("--stage", type=int, default=9, help="how much of the twin to build: 0 manual_seed only, 1 model, 2 + EMA, 3 + optimizer")
ap.add_argument("--alloc", action="store_true", help="no twin: allocate and free 8 GiB of scratch between the replays")
ap.add_argument("--twin", action="store_true", help="a second model steps eagerly between the replays (as tests/test_gpu_m2ae.py does)")
a = ap.parse_args()
torch.manual_seed(0)
model = P.PointM2AE().cuda().train()
ema = E.ModelEma(model, 0.999)
opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
args = SimpleNamespace(bf16=True, epochs=300)
pool = [make_clouds(a.batch, 2048, 100 + i, "cuda") for i in range(4)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(3):
        o = P.pretrain_step(model, ema, opt, pool[i % 4].clone(), 100, args)
        print("eager", i, {k: float(o[k]) for k in ("loss", "loss_chfr", "loss_learn", "grad_norm")})
    del o
    torch.cuda.synchronize()
    static_in = pool[0].clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = P.pretrain_step(model, ema, opt, static_in, 100, args)
if a.twin or a.twin_idle:
    torch.manual_seed(1)
    if a.stage >= 1:
        m2 = P.PointM2AE().cuda().train()
    if a.stage >= 2:
        e2 = E.ModelEma(m2, 0.999)
    if a.stage >= 3:
        o2 = E.build_optimizer(m2, lr=1e-3, flat=True, model_ema=e2)
if a.junk:
    keep, nb = [], a.junk[0]
    while nb <= a.junk[1]:
        keep += [torch.full((max(nb // 4, 1),), float("nan"), device="cuda") for _ in range(64)]
        nb *= 2
    print("junk: %d tensors" % len(keep))
for i in range(12):
    if a.twin:
        P.pretrain_step(m2, e2, o2, pool[i % 4].clone(), 100, args)
    if a.alloc:
        junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(8)]
        del junk
    static_in.copy_(pool[i % 4])
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, {k: float(out[k]) for k in ("loss", "loss_chfr", "loss_learn", "grad_norm")},
          "P finite:", bool(torch.isfinite(opt.P).all()), "G finite:", bool(torch.isfinite(opt.G).all()),
          "M finite:", bool(torch.isfinite(opt.M).all()), "V finite:", bool(torch.isfinite(opt.V).all()))
    gn = float(out["grad_norm"])
    if gn != gn or abs(gn) == float("inf"):
        offs = list(opt._offs)
        names = [n for n, _ in opt._named]
        mx = [(float(opt.G[offs[k]:offs[k + 1] if k + 1 < len(offs) else None].abs().max()), n) for k, n in enumerate(names)]
        mx.sort(reverse=True)
        print("  largest |G| entries by parameter:", mx[:5], " G numel", opt.G.numel(), "last off", offs[-1], "tail max",
              float(opt.G[offs[-1]:].abs().max()))
    if not bool(torch.isfinite(opt.G).all()):
        offs = list(opt._offs)
        names = [n for n, _ in opt._named]
        bad = [n for k, n in enumerate(names) if not bool(torch.isfinite(opt.G[offs[k]:offs[k + 1] if k + 1 < len(offs) else None]).all())]
        good = [n for n in names if n not in bad]
        print("  non-finite gradients: %d of %d parameters: %s" % (len(bad), len(names), bad))
        break
