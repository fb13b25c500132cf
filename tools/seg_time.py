"""Time the segmented 4-graph step against the single-graph step without any process group (isolates the cost of the cut)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds
E.enable_tuned_gemms()
dev = torch.device("cuda")
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
pool = [make_clouds(128, 1024, 10 + i, dev) for i in range(3)]
def build(seg):
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
    ema = E.ModelEma(m, 0.9999)
    opt = E.build_optimizer(m, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema, segment_of=E.ddp_segment if seg else None)
    return m, ema, opt
def timed(step, n=30):
    for i in range(3): step(pool[i % 3])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(pool[i % 3])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
m, ema, opt = build(False)
g = E.GraphedPretrainStep(m, ema, opt, args, pool[0], 200)
print("single graph      %.3f ms" % timed(g))
m, ema, opt = build(True)
s = E.SegmentedDDPStep(m, ema, opt, args, pool[0], 200)
print("4 graphs          %.3f ms" % timed(s))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
s.static_in.copy_(pool[0]); torch.cuda.synchronize()
evs[0].record()
for k in range(4):
    s.graphs[k].replay(); evs[k + 1].record()
torch.cuda.synchronize()
print("per graph ms:", [round(evs[k].elapsed_time(evs[k + 1]), 3) for k in range(4)])
s2 = E.SegmentedDDPStep(m, ema, opt, args, pool[0], 200, use_graphs=False, warmup_iters=0)
print("segmented eager   %.3f ms" % timed(s2, 10))
