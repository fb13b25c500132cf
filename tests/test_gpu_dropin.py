"""-m gpu: the boundary through the reference's OWN import lines and call shapes.

gm3d_amd/dropin/ goes on sys.path and the three imports of P/models_mae_learn_loss.py:24-26 are executed verbatim; then
Group.fps / Group.forward's calls (P/models_mae_learn_loss.py:926-957: furthest_point_sample, transpose -> gather_operation ->
transpose, KNN(k, transpose_mode=True)(xyz, center), idx + idx_base flat gather) and the Chamfer call sites (:188
`ChamferDistanceL2().cuda()`, :407 `self.loss_func(pred, target)`, :409-412 reshape + mean) are repeated with the reference's
argument shapes, against the fixture the reference's own model produced (tests/golden/pretrain_b2_uniform.npz: center,
neighborhood, neighborhood_org, matrix, chamfer_mean).  Bar: FPS / KNN results bit-exact, Chamfer within 1e-5 relative."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ref_imports():
    path = os.path.join(ROOT, "gm3d_amd", "dropin")
    sys.path.insert(1, path)
    try:
        # P/models_mae_learn_loss.py:24-26, verbatim
        from knn_cuda import KNN
        from pointnet2_ops import pointnet2_utils
        from extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
        import knn_cuda
        import pointnet2_ops
        import extensions.chamfer_dist
        for mod in (knn_cuda, pointnet2_ops, extensions.chamfer_dist):
            assert os.path.abspath(mod.__file__).startswith(path), mod.__file__     # resolved to the shims, nothing else
        yield KNN, pointnet2_utils, ChamferDistanceL1, ChamferDistanceL2
    finally:
        sys.path.remove(path)
        for name in [n for n in sys.modules if n.split(".")[0] in ("knn_cuda", "pointnet2_ops", "extensions")]:
            del sys.modules[name]


@pytest.mark.parametrize("case", ["b2_uniform", "b4_gaussian"])
def test_group_forward_call_shapes(ref_imports, case):
    KNN, pointnet2_utils, _, _ = ref_imports
    fx = np.load(os.path.join(GOLD, "pretrain_%s.npz" % case))
    xyz = torch.from_numpy(fx["samples"]).cuda()
    num_group, group_size = 64, 32
    knn = KNN(k=group_size, transpose_mode=True)                                    # :924
    # Group.fps (:926-933)
    fps_idx = pointnet2_utils.furthest_point_sample(xyz, num_group)
    assert fps_idx.dtype == torch.int32 and fps_idx.shape == (xyz.shape[0], num_group)
    center = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
    assert np.array_equal(center.cpu().numpy(), fx["teacher_center"])
    # Group.forward (:942-957)
    batch_size, num_points, _ = xyz.shape
    _, idx = knn(xyz, center)
    assert idx.size(1) == num_group
    assert idx.size(2) == group_size
    idx_base = torch.arange(0, batch_size, device=xyz.device).view(-1, 1, 1) * num_points
    idx = idx + idx_base
    idx = idx.view(-1)
    neighborhood = xyz.view(batch_size * num_points, -1)[idx, :]
    neighborhood = neighborhood.view(batch_size, num_group, group_size, 3).contiguous()
    neighborhood_org = neighborhood
    neighborhood = neighborhood - center.unsqueeze(2)
    assert np.array_equal(neighborhood_org.cpu().numpy(), fx["teacher_neighborhood_org"])
    assert np.array_equal(neighborhood.cpu().numpy(), fx["teacher_neighborhood"])


@pytest.mark.parametrize("case", ["b2_uniform", "b4_gaussian"])
def test_forward_loss_call_shapes(ref_imports, case):
    _, _, ChamferDistanceL1, ChamferDistanceL2 = ref_imports
    fx = np.load(os.path.join(GOLD, "pretrain_%s.npz" % case))
    loss_func = ChamferDistanceL2().cuda()                                          # :188
    Mn = int(fx["mask_num"])
    mask = torch.from_numpy(fx["mask_e200"]).bool().cuda()
    target = torch.from_numpy(fx["teacher_neighborhood"]).cuda()                    # (N, t, n, D), same samples for both passes
    pred = torch.from_numpy(fx["student_pix_pred"]).cuda()[:, -Mn:]
    N, t, n, D = target.shape
    target = target[mask].reshape(-1, n, D)                                         # :396-397
    pred = pred.reshape(-1, n, D).requires_grad_(True)
    pred32 = pred.to(dtype=torch.float32)
    target = target.to(dtype=torch.float32)
    loss = loss_func(pred32, target)                                                # :407
    loss = loss.reshape(N, -1, n)                                                   # :409
    ref = torch.from_numpy(fx["matrix"]).cuda()
    assert float((loss.mean(dim=-1) - ref).abs().max() / ref.abs().max()) <= 1e-5
    assert abs(float(loss.mean()) - float(fx["chamfer_mean"])) <= 1e-5 * float(fx["chamfer_mean"])
    loss.mean().backward()                                                          # the call site is differentiated through
    assert pred.grad is not None and bool(torch.isfinite(pred.grad).all()) and float(pred.grad.abs().sum()) > 0
    # L1 form (models/Point_MAE.py:423-424 picks it for loss == 'cdl1'): scalar, finite, differentiable
    l1 = ChamferDistanceL1().cuda()(pred32.detach().requires_grad_(True), target)
    assert l1.dim() == 0 and bool(torch.isfinite(l1))
