"""Which PyTorch (non-gm3d, non-GEMM) kernels does one pretrain step still launch, and from which source line?
python tools/leftover_ops.py   (GPU box; eager step under torch.profiler with stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds

B = int(os.environ.get("B", 128))
dev = torch.device("cuda")
torch.manual_seed(0)
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
x0 = make_clouds(B, 1024, 1, dev)
from types import SimpleNamespace
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)


def step():
    return E.pretrain_step(model, ema, opt, x0, epoch=200, args=args)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by_site = collections.defaultdict(lambda: [0, 0.0, set()])
n_kern = 0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    names = [k.name for k in ev.kernels]
    if all(("gm3d" in n or n.startswith("Cijk") or n.startswith("Custom_Cijk")) for n in names):
        continue
    site = "?"
    for fr in (ev.stack or []):
        if here in fr and "tools/leftover_ops" not in fr:
            site = fr.replace(here + "/", "")
            break
    if site == "?":
        site = str(ev.input_shapes)[:110]
    rec = by_site[(site, ev.name)]
    rec[0] += len(names)
    rec[1] += sum(k.duration for k in ev.kernels)
    n_kern += len(names)
rows = sorted(by_site.items(), key=lambda kv: -kv[1][0])
print("leftover torch kernels per step: %d" % n_kern)
for (site, op), (n, us, _) in rows[:120]:
    print("%4d %8.1f us  %-28s %s" % (n, us, op, site))
