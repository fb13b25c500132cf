import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds
E.enable_tuned_gemms()
dev = torch.device("cuda")
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
x = make_clouds(128, 1024, 10, dev)
seg = sys.argv[1] == "seg"
torch.manual_seed(0)
m = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(m, 0.9999)
opt = E.build_optimizer(m, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema, segment_of=E.ddp_segment if seg else None)
s = E.SegmentedDDPStep(m, ema, opt, args, x, 200) if seg else E.GraphedPretrainStep(m, ema, opt, args, x, 200)
for i in range(20): s(x)
torch.cuda.synchronize()
