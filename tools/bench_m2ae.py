"""Point-M2AE + GeoMask3D pretrain step (BASELINE config #4: B=128 clouds of 2048 points, bf16) on 1 GPU: eager and hipGraph replay.
    python tools/bench_m2ae.py [--batch 128] [--steps 20]   -> one JSON line"""
import argparse, json, os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, point_m2ae as P
from bench import make_clouds, roofline_of
from gm3d_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--staged", action="store_true", help="A/B: training graph + the next batch's grouping graph on a second stream")
ap.add_argument("--set", action="append", default=[], metavar="mod.ATTR=value",
                help="flip a module switch of gm3d_amd for a same-box A/B (e.g. --set gemm.WS_BN=False --set point_m2ae.VISIBLE_FIRST=False)")
ap.add_argument("--narrow-attn", action="store_true", help="A/B: four tiles per workgroup in the masked attention kernels (the round-3 form)")
a = ap.parse_args()
import importlib
for item in a.set:
    name, val = item.split("=")
    mod, attr = name.rsplit(".", 1)
    if mod == "capi":          # --set capi.gm3d_gemm_ws_set_tm2=1: a process-wide knob of the C ABI
        from gm3d_amd._capi import lib as _l
        assert getattr(_l, attr)(int(eval(val))) == 0
        continue
    setattr(importlib.import_module("gm3d_amd." + mod), attr, eval(val))
if a.narrow_attn:
    from gm3d_amd._capi import lib as _lib
    _lib.gm3d_attention_masked_set_wide(0)
# (no TunableOp table: since round 4 every product of this step runs on a hand-written kernel -- rocprof: 0 library launches)
torch.manual_seed(0)
model = P.PointM2AE().cuda().train()
ema = E.ModelEma(model, 0.999)
opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
args = SimpleNamespace(bf16=True, epochs=300)
pool = [make_clouds(a.batch, 2048, 100 + i, "cuda") for i in range(4)]


def run(step, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        o = step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, o


eager = lambda i: P.pretrain_step(model, ema, opt, pool[i % 4].clone(), 100, args)
run(eager, a.warmup)
t_eager, o = run(eager, a.steps)
line = {"metric": "point-clouds/sec Point-M2AE+GM3D pretrain step (N=2048, G=512/256/64)", "unit": "clouds/s", "n_gpus": 1,
        "dtype": "bf16", "data": "synthetic", "batch": a.batch, "eager_ms_per_step": t_eager * 1e3, "eager_clouds_per_s": a.batch / t_eager,
        "loss": float(o["loss"])}
try:
    if not a.staged:          # the whole step as ONE graph (grouping in line): what bench.py's secondary leg runs
        static_in = pool[0].clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = P.pretrain_step(model, ema, opt, static_in, 100, args)

        def replay(i):
            static_in.copy_(pool[i % 4])
            g.replay()
            return out
    else:                     # training graph + the next batch's grouping graph on a second stream (point_m2ae.GraphedM2AEStep)
        gs = P.GraphedM2AEStep(model, ema, opt, args, pool[0], 100)
        replay = lambda i: gs(pool[i % 4], next_pts=pool[(i + 1) % 4])
    run(replay, a.warmup)
    t_graph, o = run(replay, a.steps)
    gl = float(o["loss"])
    if gl != gl or abs(gl) == float("inf"):     # this model still runs PyTorch reductions that are not replay-safe on this stack
        raise RuntimeError("replayed loss is not finite (torch multi-block reductions under replay: DESIGN 3c)")
    line.update(graph_ms_per_step=t_graph * 1e3, value=a.batch / t_graph, ms_per_step=t_graph * 1e3, graph_loss=gl,
                execution="hipGraph replay")
except Exception as ex:
    line.update(value=a.batch / t_eager, ms_per_step=t_eager * 1e3, execution="eager (capture failed: %s)" % str(ex)[:100])
# per-kernel rooflines of the hand-written kernels (HIP events around every launch of two eager steps): the masked attention's
# fraction of the bf16 MFMA peak (dense flop count: blocked pairs are computed too) and the GEMM families'
probe = ops.KernelTimer()
ops.set_kernel_timer(probe)
for i in range(2):
    eager(i)
ops.set_kernel_timer(None)
roof = {}
for n, v in probe.summary().items():
    q = roofline_of(n, v["per_launch"], v["total_ms"])
    if q is None:
        continue
    roof[n] = {"bound": q["bound"], "achieved": round(q["achieved"], 2), "unit": q["unit"], "frac": round(q["frac"], 4),
               "launches_per_step": v["launches"] / 2, "ms_per_step": round(v["total_ms"] / 2, 3)}
    for k in ("tflops", "frac_mfma", "gbs", "frac_hbm"):
        if k in q:
            roof[n][k] = q[k]
line["kernel_rooflines"] = dict(sorted(roof.items(), key=lambda kv: -kv[1]["ms_per_step"])[:10])
print(json.dumps(line))
