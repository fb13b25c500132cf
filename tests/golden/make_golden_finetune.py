"""Generates tests/golden/finetune_b4.npz by running the REFERENCE's own fine-tune model
(/root/reference/Point-MAE_SA3D/models/Point_MAE.py::PointTransformer, imported in place, never copied) on CPU in
this container.  Pins the Python glue of the fine-tune path (cls token / cls_pos wiring, in-tree Block/Attention/Mlp twin,
max+cls pooling, classification head, get_loss_acc) and util/lr_decay.py's parameter groups.  The absent third-party
packages are supplied as sys.modules entries backed by the CPU oracle (FPS/KNN arithmetic stays "parity unpinned");
timm is reduced to DropPath / trunc_normal_ (the transformer itself is the reference's in-tree code), easydict to a dict
with attribute access.

Weights: oracle.model_ref.det_fill_ (values depend only on seed+name+shape), so tests rebuild them.
Usage:  python tests/golden/make_golden_finetune.py
"""
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


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def install_absent_packages():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", DropPath=R.DropPath, trunc_normal_=nn.init.trunc_normal_)
    mod("easydict", EasyDict=AttrDict)
    mod("knn_cuda", KNN=O.KNN)
    pu = mod("pointnet2_ops.pointnet2_utils", furthest_point_sample=O.furthest_point_sample,
             gather_operation=O.gather_operation)
    mod("pointnet2_ops", pointnet2_utils=pu)
    mod("extensions")
    mod("extensions.chamfer_dist", ChamferDistanceL1=O.ChamferDistanceL1, ChamferDistanceL2=O.ChamferDistanceL2)
    for name in ("matplotlib", "matplotlib.pyplot", "mpl_toolkits", "mpl_toolkits.mplot3d"):
        try:
            __import__(name)
        except Exception:
            mod(name, Axes3D=None)


def npy(t):
    return t.detach().cpu().numpy().copy()


def main():
    install_absent_packages()
    sys.path.insert(0, REF)
    from models.Point_MAE import PointTransformer  # the reference class itself
    import util.lr_decay as ref_lrd

    cfg = AttrDict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64,
                   encoder_dims=384)                               # P/cfgs/finetune_modelnet.yaml:57-67
    torch.manual_seed(0)
    ref = PointTransformer(cfg)
    R.det_fill_(ref, seed=3)
    out = {"state_keys": np.array(list(ref.state_dict().keys())),
           "state_shapes": np.array([str(list(v.shape)) for v in ref.state_dict().values()])}

    B = 4
    pts = clouds.FAMILIES["gaussian"](B, 1024, seed=4321)
    targets = torch.tensor([3, 17, 39, 0])
    out["pts"], out["targets"] = npy(pts), npy(targets)

    # eval forward
    ref.eval()
    with torch.no_grad():
        logits = ref(pts.clone())
    out["eval_logits"] = npy(logits)
    loss, acc = ref.get_loss_acc(logits, targets)
    out["eval_loss"], out["eval_acc"] = npy(loss), npy(acc)

    # train forward/backward with Dropout disabled (p=0) and the DropPath draws recorded
    ref.train()
    for m in ref.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    R._droppath_log = []
    torch.manual_seed(5)
    logits = ref(pts.clone())
    out["droppath_masks"] = npy(torch.stack(R._droppath_log))
    R._droppath_log = None
    loss = nn.functional.cross_entropy(logits, targets)
    loss.backward()
    out["train_logits"], out["train_loss"] = npy(logits), npy(loss)
    for name in ("cls_token", "cls_pos", "encoder.first_conv.0.weight", "encoder.second_conv.3.weight", "pos_embed.0.weight",
                 "blocks.blocks.0.attn.qkv.weight", "blocks.blocks.11.mlp.fc2.weight", "norm_p.weight",
                 "cls_head_finetune.0.weight", "cls_head_finetune.8.bias"):
        g = dict(ref.named_parameters())[name].grad
        out["gradnorm/" + name] = npy(g.double().norm())
        out["grad/" + name] = npy(g) if g.numel() <= 20000 else npy(g.flatten()[::7])   # big tensors: every 7th element
    out["bn_head_running_mean"] = npy(ref.cls_head_finetune[1].running_mean)

    # layer-wise lr decay groups (util/lr_decay.py) -- printed by the reference, captured here by name
    import io, contextlib, json
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        groups = ref_lrd.param_groups_lrd(ref, 0.05, no_weight_decay_list=[{"pos_embed", "cls_token"}], layer_decay=0.75)
    names = json.loads(buf.getvalue().split("parameter groups: \n", 1)[1])
    out["lrd_json"] = np.array(json.dumps({k: {"lr_scale": v["lr_scale"], "weight_decay": v["weight_decay"], "params": v["params"]}
                                           for k, v in names.items()}))
    assert len(groups) == len(names)
    np.savez_compressed(os.path.join(HERE, "finetune_b4.npz"), **out)
    print("wrote finetune_b4.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
