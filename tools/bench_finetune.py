"""Throughput of the fine-tune iteration (SURVEY.md 8f.2, P/cfgs/finetune_modelnet.yaml: 8192-point clouds, npoints 1024,
G=64, k=32, 40 classes), eager and hipGraph replay, with per-stage HIP-event times.
python tools/bench_finetune.py [--batch 32] [--steps 20] [--fp32]     (GPU box; prints one JSON line)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn
from types import SimpleNamespace
from gm3d_amd import engine_finetune as EF, engine_pretrain as E, ops
from gm3d_amd.point_transformer import PointTransformer
from bench import make_clouds, algorithmic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--points", type=int, default=8192)
ap.add_argument("--fp32", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
E.enable_tuned_gemms()
torch.manual_seed(0)
cfg = dict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64, encoder_dims=384)
model = PointTransformer(cfg).to(dev).train()
crit = nn.CrossEntropyLoss()
args = SimpleNamespace(lr=5e-4, min_lr=1e-6, warmup_epochs=10, epochs=300)
pool = [make_clouds(a.batch, a.points, 100 + i, dev) for i in range(3)]
targets = (torch.arange(a.batch, device=dev) * 7) % 40
res = {}

# ---- stages (eager, events)
def stage_times():
    marks = []
    def mark(n):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((n, e))
    opt = EF.build_optimizer(model, lr=5e-4)
    acc = {}
    for it in range(4):
        marks.clear(); mark("start")
        pts = EF.sample_points(pool[it % 3], 1024); mark("fps8192->1200+gather")
        pts = E.train_transforms(pts); mark("augment")
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not a.fp32):
            nb, c, _ = model.group_divider(pts); mark("fps64+knn")
            tok = model.encoder(nb); mark("embed")
            out = model(pts); mark("model.forward(all)")
            loss = crit(out.float(), targets)
        loss.backward(); mark("backward")
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0); opt.step(); opt.zero_grad(set_to_none=True); mark("clip+adamw")
        torch.cuda.synchronize()
        if it:
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1) / 3
    return {k: round(v, 3) for k, v in acc.items()}

res["stage_ms_eager"] = stage_times()

def timed(step):
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert np.isfinite(float(out["loss"]))
    return a.batch * a.steps / dt, dt / a.steps * 1e3

opt = EF.build_optimizer(model, lr=5e-4)
def eager(i):
    EF.adjust_learning_rate(opt, 20 + i / 100.0, args)
    return EF.finetune_step(model, crit, opt, pool[i % 3], targets, npoints=1024, max_norm=10.0, bf16=not a.fp32)
res["eager_clouds_per_s"], res["eager_ms"] = timed(eager)

probe = ops.KernelTimer()
ops.set_kernel_timer(probe)
for i in range(2):
    eager(i)
ops.set_kernel_timer(None)
ps = probe.summary()
res["hip_kernels_ms_per_step"] = {n: round(v["total_ms"] / 2, 4) for n, v in sorted(ps.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}
dom = max((n for n in ps if algorithmic(n, ps[n]["meta"])), key=lambda n: ps[n]["total_ms"])
work = sum(algorithmic(dom, m)[1] for _, m in ps[dom]["per_launch"])
b_, _, unit = algorithmic(dom, ps[dom]["meta"])
rate = work / (ps[dom]["total_ms"] * 1e-3)
res["roofline"] = {"kernel": dom, "bound": b_, "achieved": rate / (1e9 if b_ == "hbm" else 1e12), "peak": 8000.0 if b_ == "hbm" else 2500.0,
                   "unit": "GB/s" if b_ == "hbm" else "TFLOP/s", "frac": rate / (8e12 if b_ == "hbm" else 2.5e15),
                   "avg_launch_us": ps[dom]["avg_ms"] * 1e3}

try:
    gopt = EF.build_optimizer(model, lr=5e-4, capturable=True)
    g = EF.GraphedFinetuneStep(model, crit, gopt, pool[0], targets, npoints=1024, max_norm=10.0, bf16=not a.fp32)
    def graphed(i):
        EF.adjust_learning_rate(gopt, 20 + i / 100.0, args)
        return g(pool[i % 3], targets)
    res["graph_clouds_per_s"], res["graph_ms"] = timed(graphed)
except Exception as ex:
    res["graph_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:200])
try:
    fopt = EF.build_optimizer(model, lr=5e-4, flat=True, max_norm=10.0)
    gf = EF.GraphedFinetuneStep(model, crit, fopt, pool[0], targets, npoints=1024, max_norm=10.0, bf16=not a.fp32)
    def graphed_flat(i):
        EF.adjust_learning_rate(fopt, 20 + i / 100.0, args)
        return gf(pool[i % 3], targets)
    res["graph_flat_clouds_per_s"], res["graph_flat_ms"] = timed(graphed_flat)
except Exception as ex:
    res["graph_flat_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:200])
try:
    go = EF.GraphedFinetuneStep(model, crit, fopt, pool[0], targets, npoints=1024, max_norm=10.0, bf16=not a.fp32, warmup_iters=0,
                                overlap_sampling=True)
    def graphed_overlap(i):
        EF.adjust_learning_rate(fopt, 20 + i / 100.0, args)
        return go(pool[i % 3], targets, next_points=pool[(i + 1) % 3])
    res["graph_flat_overlap_clouds_per_s"], res["graph_flat_overlap_ms"] = timed(graphed_overlap)
except Exception as ex:
    res["graph_flat_overlap_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:200])
res.update(metric="point-clouds/sec fine-tune step (8192->1024 pts, G=64, cls 40)", batch=a.batch, dtype="f32" if a.fp32 else "bf16")
print(json.dumps(res))
