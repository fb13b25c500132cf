"""SegmentedDDPStep at B = 128 in ONE process (gloo group of one): the three backward segments run eagerly without collectives vs the
captured four-graph step on the same inputs -- flat gradient buffer compared by parameter name.   python tools/seg_b128_diag.py [B]"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M, fused, gemm
for item in os.environ.get("SET", "").split():        # SET="fused.ASYNC_WGRAD=False models_mae_learn_loss.PARALLEL_DECODERS=False"
    name, val = item.split("=")
    mod, attr = name.rsplit(".", 1)
    import importlib
    setattr(importlib.import_module("gm3d_amd." + mod), attr, eval(val))
    print("set", name, val)
from tests import clouds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
use_graphs = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
def poison(gb=24):
    """freed device memory full of NaN patterns, both in the caching allocator's free blocks and returned to the driver: whatever reads
    memory it never wrote shows up as NaN / a mismatch instead of passing on a fresh box's zero pages."""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    a = torch.full((gb << 28,), float("nan"), device="cuda")
    del a
    torch.cuda.empty_cache()
    a = torch.full((gb << 28,), float("nan"), device="cuda")
    del a
    torch.cuda.synchronize()


POISON = os.environ.get("POISON") == "1"
if POISON:
    poison()
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
data = clouds.gaussian(B, 1024, 900).cuda()
noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(950)).cuda()
torch.manual_seed(100)
m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
for mod in m.modules():
    if isinstance(mod, M.DropPath):
        mod.drop_prob = 0.0
ema = E.ModelEma(m, 0.999)
opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment)
E.adjust_learning_rate(opt, 200.0, args)
seg = E.SegmentedDDPStep(m, ema, opt, args, data, 200, warmup_iters=int(os.environ.get("WARM", "0")), augment=False, inject_mask_noise=True, use_graphs=use_graphs)
seg.static_noise.copy_(noise)
def eager_local():
    bufs = [t.detach().clone() for t in m.buffers()]
    seg._phase1(data)
    seg._phase2()
    seg._phase3()
    seg._cut1 = seg._cut2 = seg._cut3 = None
    with torch.no_grad():
        for t, v in zip(m.buffers(), bufs):
            t.copy_(v)
    torch.cuda.synchronize()
    return opt.G.detach().clone()
g1 = eager_local()
if POISON:
    poison()
g2 = eager_local()
print("eager vs eager equal:", torch.equal(g1, g2), float((g1 - g2).abs().max()))
P0 = opt.P.clone()
o = seg(data, noise)
torch.cuda.synchronize()
g3 = opt.G.detach().clone()
o = seg(data, noise) if False else None
print("step (%s) vs eager equal:" % ("graph" if use_graphs else "eager"), torch.equal(g1, g3), "max dev", float((g1 - g3).abs().max()), "of", float(g1.abs().max()))
offs = list(opt._offs) + [opt.G.numel()]
worst = sorted(((float((g1[o_:e] - g3[o_:e]).abs().max()) / max(float(g1[o_:e].abs().max()), 1e-12), n) for (n, _), o_, e in zip(opt._named, offs[:-1], offs[1:])), reverse=True)[:8]
for w in worst:
    print("  %.3e  %s" % w)
print("by absolute deviation (dev, max |g|, name):")
absw = sorted(((float((g1[o_:e] - g3[o_:e]).abs().max()), float(g1[o_:e].abs().max()), n) for (n, _), o_, e in zip(opt._named, offs[:-1], offs[1:])), reverse=True)[:12]
for w in absw:
    print("  %.3e  %.3e  %s" % w)
nz = sum(1 for (n, _), o_, e in zip(opt._named, offs[:-1], offs[1:]) if not torch.equal(g1[o_:e], g3[o_:e]))
print("parameters that differ at all: %d of %d" % (nz, len(opt._named)))

# a second replay on the same inputs (parameters have moved: restore them first)
