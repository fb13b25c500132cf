"""-m gpu: the product model (gm3d_amd.models_mae_learn_loss on the HIP kernels) against
 (a) the golden fixtures produced by the REFERENCE's own model file (tests/golden/make_golden.py), and
 (b) the CPU oracle on fresh seeded inputs,
in fp32.  Bar (BASELINE.json north_star): FPS/KNN-derived centres and neighbourhoods bit-exact,
Chamfer loss and encoder activations within 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b, floor=1e-12):
    """max |a-b| / max(|b|max, floor).  `floor` is an absolute scale for tensors that are analytically zero
    (e.g. the gradient of a bias that feeds a BatchNorm), where only rounding noise is left on both sides."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


@pytest.fixture(scope="module")
def M():
    from gm3d_amd import models_mae_learn_loss as M
    return M


@pytest.fixture(scope="module")
def model(M):
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b(norm_pix_loss=False)
    R.det_fill_(m, seed=0)          # same name-derived weights the fixtures were made with
    return m.cuda()


class FeedDropPath:
    """Replays the reference run's DropPath keep-masks in call order, for both execution paths of the package:
    `drop_path` (per-op modules) and `drop_path_scale` (fused stack)."""

    def __init__(self, masks):
        self.masks = [torch.from_numpy(m) for m in masks]

    def __call__(self, x, p, training):
        if p == 0.0 or not training:
            return x
        m = self.masks.pop(0).to(x.device, x.dtype).reshape((x.shape[0],) + (1,) * (x.ndim - 1))
        return x.div(1.0 - p) * m

    def scale(self, B, p, training, device):
        if p == 0.0 or not training:
            return None
        return (self.masks.pop(0).to(device, torch.float32) / (1.0 - p)).contiguous()


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("case", ["b2_uniform", "b4_gaussian"])
def test_against_reference_fixtures(M, model, case, fused, monkeypatch):
    """fused=True: the hand-written transformer-stack forward/backward (gm3d_amd/fused.py, the default);
    fused=False: the per-op PyTorch modules around the same HIP kernels."""
    monkeypatch.setattr(M, "FUSED_STACK", fused)
    fx = np.load(os.path.join(GOLD, "pretrain_%s.npz" % case))
    saved = {k: v.clone() for k, v in model.state_dict().items()}
    try:
        _fixture_case(M, model, fx, monkeypatch)
    finally:
        model.load_state_dict(saved)
        model.zero_grad()


def _fixture_case(M, model, fx, monkeypatch):
    from gm3d_amd.engine_pretrain import train_transforms
    pts = torch.from_numpy(fx["pts"]).cuda()
    samples = train_transforms(pts, draws=(torch.from_numpy(fx["scale"]), torch.from_numpy(fx["shift"])))
    assert rel(samples, fx["samples"]) <= 1e-7
    samples = torch.from_numpy(fx["samples"]).cuda()
    B = samples.shape[0]

    model.eval()
    with torch.no_grad():
        t = model(samples, mask=torch.zeros(B, 64, dtype=torch.bool, device="cuda"))
        noaug = model(samples, mask=torch.zeros(B, 64, dtype=torch.bool, device="cuda"), noaug=True)
    assert np.array_equal(t["center"].cpu().numpy(), fx["teacher_center"])                    # FPS: bit-exact
    assert np.array_equal(t["neighborhood"].cpu().numpy(), fx["teacher_neighborhood"])        # KNN: bit-exact
    assert np.array_equal(t["neighborhood_org"].cpu().numpy(), fx["teacher_neighborhood_org"])
    assert rel(t["features"], fx["teacher_features"]) <= 1e-5       # encoder activations
    assert rel(noaug, fx["teacher_noaug"]) <= 1e-5
    assert rel(t["pix_pred"], fx["teacher_pix_pred"]) <= 1e-5
    assert rel(t["loss_pred"], fx["teacher_loss_pred"]) <= 1e-5

    lp = torch.from_numpy(fx["teacher_loss_pred"]).cuda()
    m0 = model.generate_mask(lp, 0.6, epoch=0, total_epoch=400, noise=torch.from_numpy(fx["mask_e0_noise"]))
    assert np.array_equal(m0.cpu().numpy(), fx["mask_e0"])
    # guided branch: replay the reference's np.random.shuffle as a noise ranking
    rng = np.random.RandomState(int(fx["mask_e200_np_seed"]))
    noise = torch.zeros(B, 64)
    order = torch.argsort(lp.cpu(), dim=1)
    for i in range(B):
        rest = np.delete(np.arange(64), order[i, -9:].numpy())
        rng.shuffle(rest)
        noise[i, torch.from_numpy(rest)] = torch.arange(len(rest), dtype=torch.float32)
    m200 = model.generate_mask(lp, 0.6, epoch=200, total_epoch=400, noise=noise)
    assert np.array_equal(m200.cpu().numpy(), fx["mask_e200"])

    model.train()
    mask = torch.from_numpy(fx["mask_e200"]).bool().cuda()
    feed = FeedDropPath(fx["droppath_masks"])
    monkeypatch.setattr(M, "drop_path", feed)
    monkeypatch.setattr(M, "drop_path_scale", feed.scale)
    s = model(samples, mask=mask)
    assert feed.masks == []
    Mn = int(fx["mask_num"])
    assert s["mask_num"] == Mn
    assert rel(s["features"], fx["student_features"]) <= 1e-5
    assert rel(s["pix_pred"], fx["student_pix_pred"]) <= 1e-5
    assert rel(s["loss_pred"], fx["student_loss_pred"]) <= 1e-5
    lo = model.forward_loss(s["pix_pred"][:, -Mn:], s["neighborhood"], s["mask"])
    ll = model.forward_learning_loss(s["loss_pred"][:, -Mn:], mask, lo["matrix"].detach(), relative=True)
    assert rel(lo["Chamfer_mean"], fx["chamfer_mean"]) <= 1e-5      # Chamfer loss parity
    assert rel(lo["matrix"], fx["matrix"]) <= 1e-5
    assert float(lo["MSE_mean"]) == 0.0
    assert rel(ll, fx["loss_learn"]) <= 1e-5
    assert rel(model.forward_learning_loss(s["loss_pred"][:, -Mn:], mask, lo["matrix"].detach(), relative=False),
               fx["loss_learn_abs"]) <= 1e-5
    model.zero_grad()
    (13.889 * lo["MSE_mean"] + lo["Chamfer_mean"] + ll).backward()
    named = dict(model.named_parameters())
    gsq = sum(float(p.grad.double().pow(2).sum()) for p in named.values())
    assert abs(gsq ** 0.5 - float(fx["grad_norm"])) <= 1e-5 * float(fx["grad_norm"])
    for key in fx.files:
        if key.startswith("grad::"):
            g = named[key[6:]].grad
            ref = torch.from_numpy(fx[key])
            noise = 1e-5 * float(fx["grad_norm"])     # analytically-zero gradients (biases feeding a BatchNorm) carry only rounding noise
            assert rel(g[: ref.shape[0]] if g.numel() > 65536 else g, ref, floor=noise) <= 5e-5, key   # fp32 backward through 20 blocks
            assert abs(float(g.double().norm()) - float(fx["gradnorm::" + key[6:]])) <= 2e-5 * float(fx["gradnorm::" + key[6:]]) + noise
        if key.startswith("bn_after::"):
            assert rel(model.state_dict()[key[10:]], fx[key]) <= 1e-5, key


def test_step_against_oracle(M):
    """One whole optimisation step (teacher -> mask -> student -> losses -> clip -> AdamW -> EMA) on a fresh
    seeded batch: product engine on the GPU vs oracle engine on the CPU, DropPath disabled on both sides
    (its draws cannot be shared across devices), mask noise injected."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E
    from tests import clouds
    torch.manual_seed(0)
    om = R.det_fill_(R.PointMAEGM3D(drop_path_rate=0.0), seed=3)
    pm = M.mae_vit_base_patch16_dec512d8b()
    for mod in pm.modules():
        if isinstance(mod, M.DropPath):
            mod.drop_prob = 0.0
    R.det_fill_(pm, seed=3)
    pm = pm.cuda()
    om.train(); pm.train()
    oema, pema = R.ModelEma(om, decay=0.999), E.ModelEma(pm, decay=0.999)
    oopt = torch.optim.AdamW(R.param_groups(om, 0.05), lr=1e-3)
    popt = E.build_optimizer(pm, lr=1e-3, weight_decay=0.05)
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=False, accum_iter=1)
    x = clouds.gaussian(4, 1024, seed=77)
    rng = np.random.RandomState(5)
    lp_order_noise = torch.rand(4, 64, generator=torch.Generator().manual_seed(9))

    # oracle side: guided branch wants an np RandomState; to share the permutation we drive BOTH sides by
    # the same noise ranking (the oracle's generate_mask with rng=_NoiseRng below reproduces it)
    class _NoiseRng:
        def __init__(self, noise):
            self.noise, self.i = noise, 0

        def shuffle(self, arr):
            n = self.noise[self.i][torch.from_numpy(arr)]
            arr[:] = arr[torch.argsort(n).numpy()]
            self.i += 1

    pre = {k: v.detach().cpu().clone() for k, v in pm.named_parameters()}
    pre_ema = {k: v.detach().cpu().clone() for k, v in pema.ema.state_dict().items()}
    ores = R.pretrain_step(om, oema, oopt, x.clone(), epoch=200, total_epoch=400, mask_rng=_NoiseRng(lp_order_noise))
    pres = E.pretrain_step(pm, pema, popt, x.clone().cuda(), 200, args, mask_noise=lp_order_noise, augment=False)
    assert torch.equal(pres["mask"].cpu(), ores["mask"])
    assert rel(pres["loss_chfr"], ores["chamfer"]) <= 1e-5
    assert rel(pres["loss_learn"], ores["loss_learn"]) <= 1e-5
    assert rel(pres["grad_norm"], ores["grad_norm"]) <= 2e-5
    # AdamW's first update is lr*g/(|g|+eps): elements whose gradient is at rounding-noise level can flip sign
    # between two correct fp32 implementations, so the optimizer/EMA arithmetic is checked on the product's OWN
    # (clipped) gradients replayed through torch's reference AdamW on the CPU, and the gradients themselves
    # against the oracle's where they are above noise.
    cpu = {k: torch.nn.Parameter(v.clone()) for k, v in pre.items()}
    for k, p in pm.named_parameters():
        cpu[k].grad = p.grad.detach().cpu().clone()
    decay = [cpu[k] for k, p in pm.named_parameters() if not (p.dim() == 1 or k.endswith(".bias") or "token" in k)]
    nodecay = [cpu[k] for k, p in pm.named_parameters() if (p.dim() == 1 or k.endswith(".bias") or "token" in k)]
    torch.optim.AdamW([{"params": nodecay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}], lr=1e-3).step()
    psd = dict(pm.named_parameters())
    assert max(rel(psd[k], cpu[k]) for k in cpu) <= 1e-6
    og = dict(om.named_parameters())
    for k, p in pm.named_parameters():          # clipped gradients vs the oracle's clipped gradients
        # per-tensor 2e-4 relative + 1e-5 of the total gradient norm: a few first-layer gradients are small
        # differences of large BatchNorm-backward sums, i.e. fp32-noise-limited relative to the network's scale
        err = float((p.grad.double().cpu() - og[k].grad.double()).abs().max())
        assert err <= 2e-4 * float(og[k].grad.abs().max()) + 1e-5 * float(ores["grad_norm"]), k
    pesd = pema.ema.state_dict()
    for k, v in pre_ema.items():
        if v.dtype.is_floating_point:
            src = psd[k].detach().cpu() if k in psd else pm.state_dict()[k].cpu()
            assert rel(pesd[k], v * 0.999 + 0.001 * src) <= 1e-6, k


def test_bf16_step_runs_and_is_close(M):
    """Throughput mode (bf16 autocast + bf16 MFMA attention): same step, loss within bf16 tolerance of fp32."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E
    from tests import clouds
    res = {}
    for bf16 in (False, True):
        torch.manual_seed(0)
        pm = M.mae_vit_base_patch16_dec512d8b()
        R.det_fill_(pm, seed=1)
        pm = pm.cuda().train()
        for mod in pm.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
        ema = E.ModelEma(pm, decay=0.999)
        opt = E.build_optimizer(pm)
        args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=bf16, accum_iter=1)
        noise = torch.rand(8, 64, generator=torch.Generator().manual_seed(2))
        out = E.pretrain_step(pm, ema, opt, clouds.uniform(8, 1024, 5).cuda(), 0, args, mask_noise=noise, augment=False)
        res[bf16] = (float(out["loss_chfr"]), float(out["loss_learn"]), float(out["grad_norm"]))
        assert all(np.isfinite(v) for v in res[bf16])
    assert abs(res[True][0] - res[False][0]) <= 3e-2 * abs(res[False][0])   # bf16: 8 significant bits per op
    assert abs(res[True][1] - res[False][1]) <= 3e-2 * abs(res[False][1])
