"""Which aten ops (kernel-launching ones) does one pretrain step still issue from Python, and from which line of this repo?
python tools/leftover_sites.py   (GPU box; eager step under a TorchDispatchMode that records the innermost repo frame)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from types import SimpleNamespace
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_KERNEL = ("empty", "view", "reshape", "as_strided", "detach", "alias", "transpose", "permute", "expand", "slice", "select",
             "unsqueeze", "squeeze", "split", "unbind", "new_empty", "t.default", "_unsafe_view", "lift_fresh", "set_", "resize_",
             "is_", "sym_", "stride", "size", "numel", "_local_scalar", "record_stream", "_has_", "unfold", "narrow", "chunk")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        short = name.replace("aten.", "")
        if any(short.startswith(p) or ("." + p) in short for p in NO_KERNEL):
            return out
        t = out[0] if isinstance(out, (tuple, list)) and out else out
        if not (torch.is_tensor(t) and t.is_cuda):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack()):
            if fr.filename.startswith(HERE) and "tools/leftover_sites" not in fr.filename:
                site = "%s:%d" % (fr.filename[len(HERE) + 1:], fr.lineno)
                break
        self.sites[(site, short, tuple(t.shape))] += 1
        return out


dev = torch.device("cuda")
torch.manual_seed(0)
if "--m2ae" in sys.argv:          # the Point-M2AE step instead (python tools/leftover_sites.py --m2ae)
    from gm3d_amd import point_m2ae as P
    model = P.PointM2AE().to(dev).train()
    ema = E.ModelEma(model, 0.999)
    opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
    x0 = make_clouds(int(os.environ.get("B", 128)), 2048, 1, dev)
    args = SimpleNamespace(bf16=True, epochs=300)
    for _ in range(2):
        P.pretrain_step(model, ema, opt, x0.clone(), 100, args)
    torch.cuda.synchronize()
    with Sites() as s:
        P.pretrain_step(model, ema, opt, x0.clone(), 100, args)
    torch.cuda.synchronize()
    print("aten ops with a CUDA result (GEMMs included): %d" % sum(s.sites.values()))
    agg = collections.Counter()
    for (site, op, shape), n in s.sites.items():
        agg[(site, op)] += n
    for (site, op), n in sorted(agg.items(), key=lambda kv: -kv[1])[:70]:
        shapes = sorted({tuple(sh) for (st, o, sh) in s.sites if st == site and o == op})[:3]
        print("%3d  %-34s %-26s %s" % (n, site, op, shapes))
    sys.exit(0)
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
x0 = make_clouds(int(os.environ.get("B", 128)), 1024, 1, dev)
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
for _ in range(2):
    E.pretrain_step(model, ema, opt, x0, epoch=200, args=args)
torch.cuda.synchronize()
with Sites() as s:
    E.pretrain_step(model, ema, opt, x0, epoch=200, args=args)
torch.cuda.synchronize()
print("aten ops with a CUDA result (GEMMs included): %d" % sum(s.sites.values()))
for (site, op, shape), n in sorted(s.sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%3d  %-34s %-26s %s" % (n, site, op, list(shape)))
