"""Which lines of gm3d_amd still reach aten kernels in one Point-M2AE step?  A TorchDispatchMode counts every aten op of one eager step by
(op, innermost gm3d_amd source line); ops that launch nothing (views, metadata) are listed too -- read the counts beside the rocprof
kernel stats.   python tools/m2ae_aten_sites.py [--north-star]"""
import collections, os, sys, traceback
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from gm3d_amd import engine_pretrain as E
from bench import make_clouds

VIEWS = ("view", "reshape", "as_strided", "expand", "slice", "select", "transpose", "permute", "unsqueeze", "squeeze", "detach", "alias", "t.",
         "_unsafe_view", "unbind", "split", "chunk", "narrow", "size", "stride", "is_", "_local_scalar", "lift", "empty", "new_empty", "unfold",
         "record_stream", "_to_copy.default_meta", "sym_", "result_type", "set_", "resize_", "view_as")
counts = collections.Counter()


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/gm3d_amd/" in fr.filename:
                    where = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
                    break
            counts[(name.replace("aten.", ""), where)] += 1
        return func(*args, **(kwargs or {}))


torch.manual_seed(0)
if "--north-star" in sys.argv:
    from gm3d_amd import models_mae_learn_loss as M
    model = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    ema = E.ModelEma(model, 0.999)
    opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
    pts = make_clouds(128, 1024, 100, "cuda")
    step = lambda: E.pretrain_step(model, ema, opt, pts.clone(), 200, args)
else:
    from gm3d_amd import point_m2ae as P
    model = P.PointM2AE().cuda().train()
    ema = E.ModelEma(model, 0.999)
    opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
    args = SimpleNamespace(bf16=True, epochs=300)
    pts = make_clouds(128, 2048, 100, "cuda")
    step = lambda: P.pretrain_step(model, ema, opt, pts.clone(), 100, args)
for _ in range(2):
    step()
torch.cuda.synchronize()
with torch.autograd.set_multithreading_enabled(False), Sites():       # backward on this thread: the mode sees its aten calls too
    step()
torch.cuda.synchronize()
print("%d aten calls (views excluded) in one step" % sum(counts.values()))
for (op, where), c in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])):
    print("%4d  %-38s %s" % (c, op, where))
