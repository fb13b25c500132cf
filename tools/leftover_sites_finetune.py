"""Like tools/leftover_sites.py, for the fine-tune iteration (B=32 clouds of 8192 points): python tools/leftover_sites_finetune.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]]
import importlib.util, torch
import torch.nn as nn
from types import SimpleNamespace
spec = importlib.util.spec_from_file_location("ls", os.path.join(os.path.dirname(os.path.abspath(__file__)), "leftover_sites.py"))
src = open(spec.origin).read().split("dev = torch.device")[0]      # only the Sites mode, not the pretrain driver
ns = {"__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), ns)
Sites = ns["Sites"]
from gm3d_amd import engine_finetune as EF
from gm3d_amd.point_transformer import PointTransformer

torch.manual_seed(0)
model = PointTransformer(dict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64,
                              encoder_dims=384)).cuda().train()
opt = EF.build_optimizer(model, lr=5e-4, flat=True, max_norm=10.0)
crit = nn.CrossEntropyLoss()
pts = torch.randn(32, 8192, 3, device="cuda") * 0.3
tgt = (torch.arange(32, device="cuda") % 40)
for _ in range(2):
    EF.finetune_step(model, crit, opt, pts, tgt, npoints=1024, max_norm=10.0)
torch.cuda.synchronize()
with Sites() as s:
    EF.finetune_step(model, crit, opt, pts, tgt, npoints=1024, max_norm=10.0)
torch.cuda.synchronize()
print("aten ops with a CUDA result (GEMMs included): %d" % sum(s.sites.values()))
for (site, op, shape), n in sorted(s.sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%3d  %-34s %-26s %s" % (n, site, op, list(shape)))
