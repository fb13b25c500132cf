"""Extend the TunableOp table with the hipBLASLt solution for every GEMM shape of the current pretrain step that is not in it yet.
python tools/tune_gemms.py OUT.csv   (GPU box; eager steps, tuning on first use of each unknown shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tunable
from types import SimpleNamespace
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
from bench import make_clouds

out = sys.argv[1]
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gm3d_amd", "tuning", "tunableop_gfx950_b128.csv")
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(100)
tunable.read_file(src)
try:
    tunable.write_file_on_exit(False)
except AttributeError:
    pass
n0 = len(tunable.get_results())
dev = torch.device("cuda")
torch.manual_seed(0)
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
x0 = make_clouds(128, 1024, 1, dev)
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
for _ in range(2):
    E.pretrain_step(model, ema, opt, x0.clone(), epoch=200, args=args)
torch.cuda.synchronize()
# torch 2.10: results are written to `set_filename` at exit; write them explicitly in the same format as well
with open(out, "w") as f:
    for k, v in tunable.get_validators():
        f.write("Validator,%s,%s\n" % (k, v))
    for op, params, sol, t in tunable.get_results():
        f.write("%s,%s,%s,%s\n" % (op, params, sol, t))
print("entries: %d -> %d" % (n0, len(tunable.get_results())))
