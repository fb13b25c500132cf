"""Generates tests/golden/*.npz by running the REFERENCE's own model file
(/root/reference/Point-MAE_SA3D/models_mae_learn_loss.py, imported in place, never copied) on
CPU in this container.  The reference cannot travel to the GPU box; only these fixtures do.

What this pins and what it does not:
  * pinned: every line of the reference's Python glue on the pretrain path -- Group wiring,
    Encoder, pos_embed reuse, TransformerEncoder/Decoder, heads, forward_loss, generate_mask,
    forward_learning_loss, lr schedule;
  * NOT pinned: the four absent third-party packages the reference imports.  They are supplied
    here as sys.modules entries backed by the CPU oracle (oracle/ops.py, oracle/model_ref.py's
    timm-0.4.5 Block/DropPath restatement of the reference's in-tree twin models/Point_MAE.py:82-146),
    so FPS/KNN/Chamfer arithmetic stays "parity unpinned" (SURVEY.md 8c).

Weights: the reference model is constructed, then every state-dict entry is overwritten by
oracle.model_ref.det_fill_ (values depend only on seed+name+shape), so tests rebuild the identical
weights without shipping them.

Usage:  python tests/golden/make_golden.py        (writes next to this file)
"""
import json
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


def install_absent_packages():
    class PatchEmbed(nn.Module):  # only .num_patches is read (models_mae_learn_loss.py:56-57); dead weight
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("timm")
    mod("timm.models")
    mod("timm.models.vision_transformer", PatchEmbed=PatchEmbed, Block=R.Block, DropPath=R.DropPath, Mlp=R.Mlp)
    mod("knn_cuda", KNN=O.KNN)
    pu = mod("pointnet2_ops.pointnet2_utils", furthest_point_sample=O.furthest_point_sample,
             gather_operation=O.gather_operation)
    mod("pointnet2_ops", pointnet2_utils=pu)
    mod("extensions")
    mod("extensions.chamfer_dist", ChamferDistanceL1=O.ChamferDistanceL1, ChamferDistanceL2=O.ChamferDistanceL2)


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: buffers are restored in place later


def main():
    install_absent_packages()
    sys.path.insert(0, REF)
    import models_mae_learn_loss as ref_mod  # the reference file itself
    import util.lr_sched as ref_lr

    torch.manual_seed(0)
    ref = ref_mod.mae_vit_base_patch16_dec512d8b(norm_pix_loss=False, vis_mask_ratio=0.0)
    R.det_fill_(ref, seed=0)

    # ---- state-dict manifest + live-parameter census
    manifest = {k: list(v.shape) for k, v in ref.state_dict().items()}

    for name, B, fam in (("b2_uniform", 2, "uniform"), ("b4_gaussian", 4, "gaussian")):
        g = torch.Generator().manual_seed(1234)
        pts = clouds.FAMILIES[fam](B, 1024, seed=1234)
        scale = torch.rand(B, 3, generator=g) * (1.5 - 2.0 / 3.0) + 2.0 / 3.0
        shift = torch.rand(B, 3, generator=g) * 0.4 - 0.2
        samples = R.scale_and_translate_(pts.clone(), scale, shift)
        out = {"pts": npy(pts), "scale": npy(scale), "shift": npy(shift), "samples": npy(samples)}

        # -- teacher: eval mode, all-visible mask (engine_pretrain.py:86-88)
        ref.eval()
        vis = torch.zeros(B, 64, dtype=torch.bool)
        with torch.no_grad():
            t = ref(samples.clone(), mask=vis)
            feat_noaug = ref(samples.clone(), mask=vis, noaug=True)
        for k in ("pix_pred", "features", "loss_pred", "neighborhood", "neighborhood_org", "center"):
            out["teacher_" + k] = npy(t[k])
        out["teacher_noaug"] = npy(feat_noaug)

        # -- generate_mask, both branches (models_mae_learn_loss.py:744-784)
        torch.manual_seed(77)
        m0 = ref.generate_mask(t["loss_pred"], mask_ratio=0.6, guide=True, epoch=0, total_epoch=400)
        torch.manual_seed(77)
        out["mask_e0_noise"] = npy(torch.randn(B, 64))
        out["mask_e0"] = npy(m0)
        np.random.seed(99)
        m200 = ref.generate_mask(t["loss_pred"], mask_ratio=0.6, guide=True, epoch=200, total_epoch=400)
        out["mask_e200"] = npy(m200)
        out["mask_e200_np_seed"] = np.array(99)

        # -- student: train mode (BN batch stats, DropPath on), guided mask
        ref.train()
        mask = m200.flatten(1).to(torch.bool)
        R._droppath_log = []
        torch.manual_seed(5)
        bn_before = {k: v.clone() for k, v in ref.state_dict().items() if "running" in k and
                     (k.startswith("encoder.") or k.startswith("increase_dim_2."))}
        s = ref(samples.clone(), mask=mask)
        out["droppath_masks"] = npy(torch.stack(R._droppath_log))  # (n_draws, B)
        R._droppath_log = None
        M = s["mask_num"]
        lo = ref.forward_loss(s["pix_pred"][:, -M:], s["neighborhood"], s["mask"])
        ll = ref.forward_learning_loss(s["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=True)
        ll_abs = ref.forward_learning_loss(s["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=False)
        total = 13.889 * lo["MSE_mean"] + 1.0 * lo["Chamfer_mean"] + ll  # engine_pretrain.py:153,190
        ref.zero_grad()
        total.backward()
        out.update({"student_pix_pred": npy(s["pix_pred"]), "student_features": npy(s["features"]),
                    "student_loss_pred": npy(s["loss_pred"]), "mask_num": np.array(M),
                    "chamfer_mean": npy(lo["Chamfer_mean"]), "mse_mean": npy(lo["MSE_mean"]),
                    "matrix": npy(lo["matrix"]), "loss_learn": npy(ll), "loss_learn_abs": npy(ll_abs),
                    "total": npy(total)})
        live = [k for k, p in ref.named_parameters() if p.grad is not None]
        gsq = sum(float(p.grad.double().pow(2).sum()) for _, p in ref.named_parameters() if p.grad is not None)
        out["grad_norm"] = np.array(gsq ** 0.5)
        for k in ("encoder.first_conv.0.weight", "encoder.second_conv.3.bias", "pos_embed.0.weight",
                  "blocks.blocks.0.attn.qkv.weight", "blocks.blocks.11.mlp.fc2.bias", "norm_p.weight",
                  "MAE_decoder.blocks.3.attn.proj.weight", "MAE_decoder_loss_pred.norm.bias", "mask_token",
                  "increase_dim_2.3.bias", "increase_dim_just_network_without_feature.0.weight"):
            gk = dict(ref.named_parameters())[k].grad
            out["gradnorm::" + k] = np.array(float(gk.double().norm()))
            out["grad::" + k] = npy(gk if gk.numel() <= 65536 else gk[:24])  # big tensors: first 24 rows + norm
        for k in bn_before:
            out["bn_after::" + k] = npy(ref.state_dict()[k])
        # restore BN buffers so the next case starts from the same state
        with torch.no_grad():
            for k, v in bn_before.items():
                ref.state_dict()[k].copy_(v)
            for k, v in ref.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    v.zero_()
        np.savez_compressed(os.path.join(HERE, "pretrain_%s.npz" % name), **out)
        print(name, "chamfer", float(lo["Chamfer_mean"]), "learn", float(ll), "M", M, "live tensors", len(live))

    manifest_out = {"state_dict": manifest, "live_parameters": live,
                    "n_parameters": int(sum(p.numel() for p in ref.parameters())),
                    "n_live_parameters": int(sum(p.numel() for k, p in ref.named_parameters() if k in set(live)))}
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest_out, f, indent=0, sort_keys=True)

    # ---- lr schedule samples (util/lr_sched.py:11-23), driver defaults main_pretrain_multi_gpu.py:76,123-129
    class A:
        lr, min_lr, warmup_epochs, epochs = 1e-3, 0.0, 40, 400

    class FakeOpt:
        param_groups = [{"lr": 0.0}]

    eps = np.array([0.0, 0.5, 10.0, 39.99, 40.0, 41.5, 200.0, 399.0, 399.99])
    lrs = np.array([ref_lr.adjust_learning_rate(FakeOpt, float(e), A) for e in eps])
    np.savez(os.path.join(HERE, "lr_sched.npz"), epochs=eps, lrs=lrs)
    print("manifest: %d keys, %d params, %d live" % (len(manifest), manifest_out["n_parameters"],
                                                       manifest_out["n_live_parameters"]))


if __name__ == "__main__":
    main()
