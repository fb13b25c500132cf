"""Every GEMM of one pretrain step with its shape, layout, time and TFLOP/s (torch.profiler, eager step).
python tools/gemm_profile.py   (GPU box)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from torch.profiler import profile, ProfilerActivity
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds

B = int(os.environ.get("B", 128))
dev = torch.device("cuda")
E.enable_tuned_gemms()
torch.manual_seed(0)
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
x0 = make_clouds(B, 1024, 1, dev)
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
step = lambda: E.pretrain_step(model, ema, opt, x0.clone(), epoch=200, args=args)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    if ev.name not in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::matmul", "aten::linear"):
        continue
    shp = [s for s in ev.input_shapes if s]
    try:
        if ev.name == "aten::addmm":
            a, b = shp[-2], shp[-1]
        else:
            a, b = shp[0], shp[1]
        if len(a) == 3:
            flops = 2.0 * a[0] * a[1] * a[2] * b[2]
        else:
            flops = 2.0 * a[0] * a[1] * b[1]
    except Exception:
        continue
    us = sum(k.duration for k in ev.kernels)
    key = (ev.name, str(a), str(b), ev.kernels[0].name[:60])
    r = rows[key]
    r[0] += 1; r[1] += us; r[2] += flops
tot_us = sum(r[1] for r in rows.values()); tot_f = sum(r[2] for r in rows.values())
print("GEMMs: %.2f ms, %.1f GFLOP, %.0f TFLOP/s average" % (tot_us / 1e3, tot_f / 1e9, tot_f / tot_us / 1e6))
for key, (n, us, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%3d x %8.1f us  %6.0f TF/s  %-12s %-18s %-18s %s" % (n, us / n, fl / us / 1e6, key[0], key[1], key[2], key[3]))
