"""Which GEMM shapes does the Point-M2AE step run on which entry point, and how long do they take (HIP-event brackets, eager)?
python tools/m2ae_gemm_shapes.py   (GPU box)"""
import collections, os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, point_m2ae as P, ops
from bench import make_clouds

torch.manual_seed(0)
model = P.PointM2AE().cuda().train()
ema = E.ModelEma(model, 0.999)
opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
args = SimpleNamespace(bf16=True, epochs=300)
x = make_clouds(128, 2048, 1, "cuda")
for _ in range(3):
    P.pretrain_step(model, ema, opt, x.clone(), 100, args)
probe = ops.KernelTimer()
ops.set_kernel_timer(probe)
for _ in range(2):
    P.pretrain_step(model, ema, opt, x.clone(), 100, args)
ops.set_kernel_timer(None)
agg = collections.defaultdict(lambda: [0, 0.0])
for name, v in probe.summary().items():
    for ms, meta in v["per_launch"]:
        key = (name, tuple(sorted((k, str(val)) for k, val in meta.items() if k != "dtype")))
        agg[key][0] += 1
        agg[key][1] += ms
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, meta), (n, ms) in rows[:int(os.environ.get("TOP", 70))]:
    print("%-34s %-60s %5.1f/step %8.1f us/step %7.1f us each" % (name, " ".join("%s=%s" % kv for kv in meta), n / 2, ms / 2 * 1e3, ms / n * 1e3))
