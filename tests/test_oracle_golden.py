"""CPU: the oracle's restatement of the reference glue (oracle/model_ref.py) against fixtures produced by
running the REFERENCE's own models_mae_learn_loss.py (tests/golden/make_golden.py, this container only).
This is what pins the oracle for rows a3-a12 of SURVEY.md 8a; FPS/KNN/Chamfer arithmetic underneath both
sides is the same C oracle and stays "parity unpinned"."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def close(a, b, rtol=1e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return (a - b).abs().max() <= rtol * b.abs().max().clamp_min(1e-12)


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    return R.det_fill_(R.PointMAEGM3D(), seed=0)


@pytest.mark.parametrize("case", ["b2_uniform", "b4_gaussian"])
def test_oracle_reproduces_reference(model, case):
    fx = np.load(os.path.join(GOLD, "pretrain_%s.npz" % case))
    saved = {k: v.clone() for k, v in model.state_dict().items()}
    samples = torch.from_numpy(fx["samples"])
    B = samples.shape[0]
    aug = R.scale_and_translate_(torch.from_numpy(fx["pts"]).clone(), torch.from_numpy(fx["scale"]),
                                 torch.from_numpy(fx["shift"]))
    assert torch.equal(aug, samples)
    model.eval()
    with torch.no_grad():
        t = model(samples.clone(), mask=torch.zeros(B, 64, dtype=torch.bool))
    assert np.array_equal(t["center"].numpy(), fx["teacher_center"])
    assert np.array_equal(t["neighborhood"].numpy(), fx["teacher_neighborhood"])
    for k in ("pix_pred", "features", "loss_pred"):
        assert close(t[k], fx["teacher_" + k]), k
    lp = torch.from_numpy(fx["teacher_loss_pred"])
    m0 = model.generate_mask(lp, 0.6, epoch=0, total_epoch=400, noise=torch.from_numpy(fx["mask_e0_noise"]))
    assert np.array_equal(m0.numpy(), fx["mask_e0"])
    m200 = model.generate_mask(lp, 0.6, epoch=200, total_epoch=400, rng=np.random.RandomState(int(fx["mask_e200_np_seed"])))
    assert np.array_equal(m200.numpy(), fx["mask_e200"])
    assert (m200.sum(1) == 39).all()

    model.train()
    mask = torch.from_numpy(fx["mask_e200"]).bool()
    R._droppath_feed = [torch.from_numpy(r) for r in fx["droppath_masks"]]
    try:
        s = model(samples.clone(), mask=mask)
    finally:
        assert R._droppath_feed == []
        R._droppath_feed = None
    M = int(fx["mask_num"])
    assert s["mask_num"] == M == 39
    for k in ("pix_pred", "features", "loss_pred"):
        assert close(s[k], fx["student_" + k]), k
    lo = model.forward_loss(s["pix_pred"][:, -M:], s["neighborhood"], s["mask"])
    ll = model.forward_learning_loss(s["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=True)
    assert close(lo["Chamfer_mean"], fx["chamfer_mean"]) and close(lo["matrix"], fx["matrix"])
    assert float(lo["MSE_mean"]) == 0.0
    assert close(ll, fx["loss_learn"])
    assert close(model.forward_learning_loss(s["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=False),
                 fx["loss_learn_abs"])
    model.zero_grad()
    (13.889 * lo["MSE_mean"] + lo["Chamfer_mean"] + ll).backward()
    named = dict(model.named_parameters())
    gsq = sum(float(p.grad.double().pow(2).sum()) for p in named.values())
    assert abs(gsq ** 0.5 - float(fx["grad_norm"])) <= 1e-5 * float(fx["grad_norm"])
    for key in fx.files:
        if key.startswith("grad::"):
            g = named[key[6:]].grad
            ref = torch.from_numpy(fx[key])
            assert close(g[: ref.shape[0]] if g.numel() > 65536 else g, ref, rtol=2e-5), key
        if key.startswith("bn_after::"):
            assert close(model.state_dict()[key[10:]], fx[key]), key
    model.load_state_dict(saved)


def test_manifest_and_live_census(model):
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    assert man["n_parameters"] == 90475168 and man["n_live_parameters"] == 36840288
    assert len(man["state_dict"]) == 485
    assert set(k for k, _ in model.named_parameters()) == set(man["live_parameters"])
    for k, v in model.state_dict().items():
        assert list(v.shape) == man["state_dict"][k], k


def test_lr_schedule():
    fx = np.load(os.path.join(GOLD, "lr_sched.npz"))
    got = [R.adjust_learning_rate(float(e), 1e-3, 0.0, 40, 400) for e in fx["epochs"]]
    assert np.allclose(got, fx["lrs"], rtol=1e-12, atol=0)


# ---- FPS against the reference's own NumPy statement (tests/golden/make_golden_fps.py; P/datasets/ModelNetDataset.py:25-46) ----
def _fps_cases():
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "fps_modelnet.npz"))
    names = sorted({k.split("/")[0] for k in fx.files})
    return fx, names


def test_oracle_fps_matches_reference_numpy_fps(oracle_ops):
    """oracle_fps (start 0, skip rule vacuous on these clouds, fp32 (dx*dx+dy*dy)+dz*dz, first maximum) selects exactly the
    points the reference's farthest_point_sample returned -- including the `_tight` cases, where the reference's float64 run
    differs from its float32 run, i.e. only the fp32 rounding contract reproduces the selection."""
    fx, names = _fps_cases()
    assert len(names) >= 5
    for n in names:
        xyz, idx, pts = fx[n + "/xyz"], fx[n + "/idx"], fx[n + "/points"]
        got = oracle_ops.furthest_point_sample(torch.from_numpy(xyz)[None], len(idx)).numpy()[0]
        assert np.array_equal(got, idx), n
        assert np.array_equal(xyz[got], pts), n
