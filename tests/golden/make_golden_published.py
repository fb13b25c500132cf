"""Generates tests/golden/published_b4.npz by running the REFERENCE's own published-run model
(/root/reference/Point-MAE_SA3D/models_mae_learn_loss_Classifier_SVM_feature_besed.py), its frozen teacher
(models/Point_MAE.py::Point_MAE with config_m.yaml's model section) and engine_pretrain_Classifier_SVM.py's
forward_features_Decoder, imported in place on CPU in this container.  Absent third-party packages are supplied as
sys.modules entries (oracle-backed FPS/KNN/Chamfer = "parity unpinned"; timm Block/DropPath from the oracle's restatement of
the reference's in-tree twin; torchvision.transforms.Compose, torch._six.inf, easydict as trivial stand-ins; the
reference's own datasets/data_transforms.py is loaded by path because the name `datasets` is taken by another package).

Weights: oracle.model_ref.det_fill_ (seed+name+shape).   Usage:  python tests/golden/make_golden_published.py
"""
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Point-MAE_SA3D"
sys.path.insert(0, ROOT)

from oracle import model_ref as R  # noqa: E402
from oracle import ops as O  # noqa: E402
from tests import clouds  # noqa: E402
from tests.golden.make_golden_finetune import AttrDict, npy  # noqa: E402
from tests.golden import make_golden, make_golden_finetune  # noqa: E402


def install():
    make_golden.install_absent_packages()              # timm.models.vision_transformer, knn_cuda, pointnet2_ops, extensions
    sys.modules["timm.models.layers"] = types.ModuleType("timm.models.layers")
    sys.modules["timm.models.layers"].DropPath = R.DropPath
    sys.modules["timm.models.layers"].trunc_normal_ = nn.init.trunc_normal_
    sys.modules["easydict"] = types.ModuleType("easydict")
    sys.modules["easydict"].EasyDict = AttrDict
    six = types.ModuleType("torch._six")
    six.inf = math.inf
    sys.modules["torch._six"] = six
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    tvt.Compose = Compose
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    for name in ("matplotlib", "matplotlib.pyplot", "mpl_toolkits", "mpl_toolkits.mplot3d"):
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            m.Axes3D = None
            sys.modules[name] = m
    ds = types.ModuleType("datasets")
    ds.__path__ = []
    spec = importlib.util.spec_from_file_location("datasets.data_transforms", os.path.join(REF, "datasets", "data_transforms.py"))
    dt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dt)
    ds.data_transforms = dt
    sys.modules["datasets"], sys.modules["datasets.data_transforms"] = ds, dt


def main():
    install()
    sys.path.insert(0, REF)
    import models_mae_learn_loss_Classifier_SVM_feature_besed as ref_mod
    from models.Point_MAE import Point_MAE
    import engine_pretrain_Classifier_SVM as ref_eng

    torch.manual_seed(0)
    student = ref_mod.mae_vit_base_patch16_dec512d8b(norm_pix_loss=False)
    R.det_fill_(student, seed=11)
    cfg = json.load(open(os.path.join(REF, "config_m.yaml")))["model"]
    cfg = AttrDict({**cfg, "transformer_config": AttrDict(cfg["transformer_config"])})
    teacher = Point_MAE(cfg)
    R.det_fill_(teacher, seed=12)
    teacher.eval()
    out = {"student_keys": np.array(list(student.state_dict().keys())),
           "student_shapes": np.array([str(list(v.shape)) for v in student.state_dict().values()]),
           "teacher_keys": np.array(list(teacher.state_dict().keys())),
           "teacher_shapes": np.array([str(list(v.shape)) for v in teacher.state_dict().values()])}

    B = 4
    pts = clouds.FAMILIES["gaussian"](B, 1024, seed=2468)
    out["pts"] = npy(pts)
    # EMA-teacher role: the same weights in eval mode, all-visible mask (engine :119-124)
    student.eval()
    vis = torch.zeros(B, 64, dtype=torch.bool)
    with torch.no_grad():
        t = student(pts.clone(), mask=vis)
    out["ema_loss_pred"], out["ema_features"], out["ema_pix_pred"] = npy(t["loss_pred"]), npy(t["features"]), npy(t["pix_pred"])
    for epoch, after200 in ((0, False), (150, False), (299, False), (60, True)):
        np.random.seed(7 + epoch)
        torch.manual_seed(77 + epoch)
        out["mask_noise_e%d" % epoch] = npy(torch.randn(B, 64))      # what the unguided branch (:1078-1080) draws
        torch.manual_seed(77 + epoch)
        m = student.generate_mask(t["loss_pred"], mask_ratio=0.6, guide=True, epoch=epoch, total_epoch=300, after_200_epoch=after200)
        out["mask_e%d_%d" % (epoch, int(after200))] = npy(m)
    mask = torch.from_numpy(out["mask_e150_0"]).bool()

    student.train()
    R._droppath_log = []
    torch.manual_seed(5)
    s = student(pts.clone(), mask=mask)
    out["droppath_masks"] = npy(torch.stack(R._droppath_log))
    R._droppath_log = None
    M = s["mask_num"]
    with torch.no_grad():
        ft, pt, pr = ref_eng.forward_features_Decoder(teacher, t["neighborhood"], t["center"], "dino", s["pix_pred"][:, -M:], s["mask"])
    lo = student.forward_loss(s["pix_pred"][:, -M:], ft.detach(), s["mask"], pt, pr)
    ll = student.forward_learning_loss(s["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=True)
    out["mask_num"] = np.array(M)
    for k, v in (("student_features", s["features"]), ("student_pix_pred", s["pix_pred"]), ("student_loss_pred", s["loss_pred"]),
                 ("feature_target", ft), ("point_target", pt), ("point_reconstructed", pr), ("mse_mean", lo["MSE_mean"]),
                 ("chamfer_mean", lo["Chamfer_mean"]), ("matrix", lo["matrix"]), ("loss_learn", ll)):
        out[k] = npy(v)
    (13.889 * lo["MSE_mean"] + 1000.0 * lo["Chamfer_mean"] + ll).backward()
    named = dict(student.named_parameters())
    out["grad_norm"] = npy(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in named.values() if p.grad is not None)))
    out["no_grad_params"] = np.array([k for k, p in named.items() if p.grad is None])
    for name in ("mask_token", "mask_token_loss_pred", "MAE_encoder.encoder.first_conv.0.weight", "MAE_encoder.pos_embed.0.weight",
                 "MAE_encoder.blocks.blocks.3.attn.qkv.weight", "MAE_decoder.blocks.2.mlp.fc1.weight",
                 "MAE_decoder_loss_pred.blocks.11.attn.proj.weight", "decoder_pos_embed.2.weight", "increase_dim_2.0.weight",
                 "increase_dim_2.3.bias", "MAE_encoder.norm_p.weight"):
        g = named[name].grad
        out["gradnorm/" + name] = npy(g.double().norm())
        out["grad/" + name] = npy(g) if g.numel() <= 20000 else npy(g.flatten()[::7])
    np.savez_compressed(os.path.join(HERE, "published_b4.npz"), **out)
    print("wrote published_b4.npz", len(out), "entries; no-grad params:", list(out["no_grad_params"]))


if __name__ == "__main__":
    main()
