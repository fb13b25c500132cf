"""Throughput of the published-run iteration (EMA teacher + student with a 12-block loss-prediction decoder +
frozen Point-MAE teacher), B clouds of 1024 points, bf16, hipGraph replay vs eager.   python tools/bench_published.py"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from gm3d_amd import engine_pretrain_Classifier_SVM as EV, engine_pretrain as E
from gm3d_amd import models_mae_learn_loss_Classifier_SVM_feature_besed as V
from gm3d_amd.point_mae import Point_MAE
from bench import make_clouds

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda")
E.enable_tuned_gemms()
torch.manual_seed(0)
model = V.mae_vit_base_patch16_dec512d8b().to(dev).train()
teacher = Point_MAE({"group_size": 32, "num_group": 64, "loss": "cdl2",
                     "transformer_config": {"mask_ratio": 0, "mask_type": "rand", "trans_dim": 384, "encoder_dims": 384, "depth": 12,
                                            "drop_path_rate": 0.1, "num_heads": 6, "decoder_depth": 4, "decoder_num_heads": 6}}).to(dev).eval()
for p in teacher.parameters():
    p.requires_grad_(False)
ema = E.ModelEma(model, E.ema_decay_for_epoch(150))
opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
args = SimpleNamespace(mask_ratio=0.6, epochs=300, relative=True, bf16=True, accum_iter=1, after_epoch=15, loss_multiply_by=(13.889, 1000.0),
                       after_200_epoch=False, shared_learnable_tokens=False, lr=1e-3, min_lr=0.0, warmup_epochs=10)
pool = [make_clouds(a.batch, 1024, 500 + i, dev) for i in range(3)]


def timed(step):
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    v = float(out["loss"] + out["loss_learn"])
    assert v == v
    return a.batch * a.steps / dt, dt / a.steps * 1e3


res = {"metric": "point-clouds/sec published-run pretrain step (N=1024,G=64)", "batch": a.batch, "dtype": "bf16"}
res["eager_clouds_per_s"], res["eager_ms"] = timed(lambda i: EV.pretrain_step(model, ema, teacher, opt, pool[i % 3].clone(), 150, args))
try:
    g = EV.graphed_step(model, ema, teacher, opt, args, pool[0], 150)
    res["graph_clouds_per_s"], res["graph_ms"] = timed(lambda i: g(pool[i % 3]))
except Exception as ex:
    res["graph_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:200])
print(json.dumps(res))
