"""SVM-validation path: CPU checks of the pooling/SVM/gather glue; GPU check of the feature extractor
against the reference-made fixture."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pool_and_svm_match_reference_formula():
    from gm3d_amd import validate as V
    rng = np.random.RandomState(0)
    cls = rng.randint(0, 4, 300)
    feats = rng.randn(300, 64, 16).astype(np.float32) + cls[:, None, None] * 0.8
    tr, te = slice(0, 200), slice(200, 300)
    acc = V.evaluate_svm(feats[tr], cls[tr], feats[te], cls[te])
    from sklearn.svm import SVC
    clf = SVC(C=0.01, kernel="linear").fit(feats[tr].mean(1) + feats[tr].max(1), cls[tr])
    ref = (clf.predict(feats[te].mean(1) + feats[te].max(1)) == cls[te]).mean()
    assert acc == ref and acc > 0.9
    pooled = V.pool_features(torch.from_numpy(feats)).numpy()
    assert np.allclose(pooled, feats.mean(1) + feats.max(1), atol=1e-6)
    assert V.evaluate_svm(pooled[tr], cls[tr], pooled[te], cls[te]) == acc       # pooled-before-gather form
    assert V.gather_tensor(torch.ones(3)) .shape == (3,)                          # no process group: identity


@pytest.mark.gpu
def test_extract_features_matches_reference_fixture():
    from gm3d_amd import models_mae_learn_loss as M
    from gm3d_amd import validate as V
    from oracle import model_ref as R
    fx = np.load(os.path.join(GOLD, "pretrain_b2_uniform.npz"))
    torch.manual_seed(0)
    m = R.det_fill_(M.mae_vit_base_patch16_dec512d8b(), seed=0).cuda().eval()
    samples = torch.from_numpy(fx["samples"]).cuda()
    f = V.extract_features(m, samples, npoints=1024)      # FPS 1024 -> 1024 re-orders the points, not the set
    ref = torch.from_numpy(fx["teacher_noaug"])
    assert float((f.cpu() - ref).abs().max() / ref.abs().max()) <= 1e-5
    big = torch.cat([samples, samples * 0.999, samples * 1.001, samples[:, :512]], dim=1)   # 3584 points
    g = V.extract_features(m, big, npoints=1024)
    assert g.shape == (2, 64, 384) and torch.isfinite(g).all()
    # sampled set == oracle FPS on the same cloud
    from oracle import ops as O
    idx = O.furthest_point_sample(big.cpu(), 1024).long()
    exp = torch.gather(big.cpu(), 1, idx.unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(V.fps(big, 1024).cpu(), exp)
