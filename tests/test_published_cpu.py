"""CPU: published-run variant -- oracle/published_ref.py against the fixture produced by running the
REFERENCE's own models_mae_learn_loss_Classifier_SVM_feature_besed.py + models/Point_MAE.py::Point_MAE +
engine_pretrain_Classifier_SVM.forward_features_Decoder (tests/golden/make_golden_published.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as R
from oracle import published_ref as PR

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def close(a, b, rtol=1e-5, floor=1e-12):
    a, b = torch.as_tensor(a).detach().double(), torch.as_tensor(b).detach().double()
    return float((a - b).abs().max()) <= rtol * max(float(b.abs().max()), floor)


def picked(g):
    return g if g.numel() <= 20000 else g.flatten()[::7]


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLD, "published_b4.npz"))


def test_oracle_reproduces_reference_published_run(fx):
    torch.manual_seed(0)
    student = R.det_fill_(PR.PublishedGM3D(), seed=11)
    teacher = R.det_fill_(PR.FrozenPointMAE(), seed=12).eval()
    assert {k: str(list(v.shape)) for k, v in student.state_dict().items()} == dict(zip(map(str, fx["student_keys"]), map(str, fx["student_shapes"])))
    assert {k: str(list(v.shape)) for k, v in teacher.state_dict().items()} == dict(zip(map(str, fx["teacher_keys"]), map(str, fx["teacher_shapes"])))
    pts = torch.from_numpy(fx["pts"])
    B = pts.shape[0]
    student.eval()
    with torch.no_grad():
        t = student(pts.clone(), mask=torch.zeros(B, 64, dtype=torch.bool))
    for k in ("loss_pred", "features", "pix_pred"):
        assert close(t[k], fx["ema_" + k]), k
    lp = torch.from_numpy(fx["ema_loss_pred"])
    for epoch, after200 in ((0, False), (150, False), (299, False), (60, True)):
        m = student.generate_mask(lp, 0.6, True, epoch, 300, after200, rng=np.random.RandomState(7 + epoch),
                                  noise=torch.from_numpy(fx["mask_noise_e%d" % epoch]))
        assert np.array_equal(m.numpy(), fx["mask_e%d_%d" % (epoch, int(after200))]), (epoch, after200)
        assert (m.sum(1) == 39).all()
    mask = torch.from_numpy(fx["mask_e150_0"]).bool()
    student.train()
    R._droppath_feed = [torch.from_numpy(r) for r in fx["droppath_masks"]]
    try:
        s = student(pts.clone(), mask=mask)
    finally:
        assert R._droppath_feed == []
        R._droppath_feed = None
    M = s["mask_num"]
    assert M == int(fx["mask_num"]) == 39
    for k in ("features", "pix_pred", "loss_pred"):
        assert close(s[k], fx["student_" + k]), k
    ft, pt, pr = teacher.features_decoder(t["neighborhood"], t["center"], s["pix_pred"][:, -M:].detach(), s["mask"])
    assert close(ft, fx["feature_target"]) and close(pt, fx["point_target"]) and close(pr, fx["point_reconstructed"])
    lo = student.forward_loss(s["pix_pred"][:, -M:], ft, s["mask"], pt, pr)
    ll = student.forward_learning_loss(s["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=True)
    assert close(lo["MSE_mean"], fx["mse_mean"]) and close(lo["Chamfer_mean"], fx["chamfer_mean"])
    assert close(lo["matrix"], fx["matrix"]) and close(ll, fx["loss_learn"])
    (13.889 * lo["MSE_mean"] + 1000.0 * lo["Chamfer_mean"] + ll).backward()
    named = dict(student.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == sorted(map(str, fx["no_grad_params"]))
    gn = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in named.values() if p.grad is not None)))
    assert abs(gn - float(fx["grad_norm"])) <= 1e-5 * float(fx["grad_norm"])
    for k in fx.files:
        if k.startswith("grad/"):
            g = named[k[5:]].grad
            assert close(g.double().norm(), fx["gradnorm/" + k[5:]], rtol=2e-5, floor=1e-5 * gn), k
            assert float((picked(g).double() - torch.from_numpy(fx[k]).double()).abs().max()) <= 2e-5 * max(float(fx["gradnorm/" + k[5:]]), 1e-5 * gn), k


def test_product_model_keys_match_reference(fx):
    """gm3d_amd's variant model and frozen teacher carry exactly the reference's state-dict keys and shapes."""
    from gm3d_amd import models_mae_learn_loss_Classifier_SVM_feature_besed as V
    from gm3d_amd.point_mae import Point_MAE
    m = V.mae_vit_base_patch16_dec512d8b()
    assert {k: str(list(v.shape)) for k, v in m.state_dict().items()} == dict(zip(map(str, fx["student_keys"]), map(str, fx["student_shapes"])))
    cfg = {"group_size": 32, "num_group": 64, "loss": "cdl2",
           "transformer_config": {"mask_ratio": 0, "mask_type": "rand", "trans_dim": 384, "encoder_dims": 384, "depth": 12,
                                  "drop_path_rate": 0.1, "num_heads": 6, "decoder_depth": 4, "decoder_num_heads": 6}}
    t = Point_MAE(cfg)
    assert {k: str(list(v.shape)) for k, v in t.state_dict().items()} == dict(zip(map(str, fx["teacher_keys"]), map(str, fx["teacher_shapes"])))
    assert V.MaskedAutoencoderViT.keep_ratio(True, 149, 300, False) == pytest.approx(0.4)
    assert V.MaskedAutoencoderViT.keep_ratio(True, 299, 300, True) == 0.5 and V.MaskedAutoencoderViT.keep_ratio(False, 0, 300, False) == 0.5
