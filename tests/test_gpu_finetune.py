"""-m gpu: the fine-tune path -- gm3d_amd.point_transformer / engine_finetune on the HIP kernels
against (a) the fixture produced by the REFERENCE's own models/Point_MAE.py::PointTransformer and (b) the CPU oracle
(oracle/finetune_ref.py) on fresh seeded inputs, in fp32; bf16 as a sanity band.  Bar: FPS-selected points
bit-exact, logits / loss within 1e-5 relative, gradients within 5e-5 of the gradient norm."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import finetune_ref as FR
from oracle import model_ref as R
from tests import clouds
from tests.test_gpu_model import FeedDropPath, rel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64, encoder_dims=384)


@pytest.fixture(scope="module")
def M():
    from gm3d_amd import models_mae_learn_loss as M
    return M


def build(seed=3, drop_path=0.1):
    from gm3d_amd.point_transformer import PointTransformer
    torch.manual_seed(0)
    m = PointTransformer(dict(CFG, drop_path_rate=drop_path))
    R.det_fill_(m, seed=seed)
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    return m.cuda()


def picked(g):
    return g if g.numel() <= 20000 else g.flatten()[::7]


@pytest.mark.parametrize("fused", [True, False])
def test_against_reference_fixture(M, fused, monkeypatch):
    monkeypatch.setattr(M, "FUSED_STACK", fused)
    fx = np.load(os.path.join(GOLD, "finetune_b4.npz"))
    m = build()
    assert {k: str(list(v.shape)) for k, v in m.state_dict().items()} == dict(zip(map(str, fx["state_keys"]), map(str, fx["state_shapes"])))
    pts, targets = torch.from_numpy(fx["pts"]).cuda(), torch.from_numpy(fx["targets"]).cuda()
    m.eval()
    with torch.no_grad():
        logits = m(pts.clone())
    assert rel(logits, fx["eval_logits"]) <= 1e-5
    loss, acc = m.get_loss_acc(logits, targets)
    assert rel(loss, fx["eval_loss"]) <= 1e-5 and float(acc) == float(fx["eval_acc"])
    m.train()
    feed = FeedDropPath(fx["droppath_masks"])
    monkeypatch.setattr(M, "drop_path", feed)
    monkeypatch.setattr(M, "drop_path_scale", feed.scale)
    logits = m(pts.clone())
    assert feed.masks == []
    assert rel(logits, fx["train_logits"]) <= 1e-5
    loss = nn.functional.cross_entropy(logits, targets)
    assert rel(loss, fx["train_loss"]) <= 1e-5
    loss.backward()
    params = dict(m.named_parameters())
    for k in fx.files:
        if k.startswith("grad/"):
            g = params[k[5:]].grad
            gn = float(fx["gradnorm/" + k[5:]])
            assert abs(float(g.double().norm()) - gn) <= 5e-5 * gn + 1e-9, k
            assert float((picked(g).double().cpu() - torch.from_numpy(fx[k]).double()).abs().max()) <= 5e-5 * gn + 1e-9, k
    assert rel(m.cls_head_finetune[1].running_mean, fx["bn_head_running_mean"]) <= 1e-5


def test_sample_points_against_oracle(oracle_ops):
    """FPS 8192 -> 1200, shared random 1024-subset, gather (P/engine_finetune.py:117-134): bit-exact."""
    from gm3d_amd import engine_finetune as EF
    pts = clouds.gaussian(3, 8192, seed=11)
    subset = np.random.RandomState(0).choice(1200, 1024, False)
    got = EF.sample_points(pts.cuda(), 1024, subset=subset)
    want = FR.sample_points(pts, 1024, subset)
    assert got.shape == (3, 1024, 3) and torch.equal(got.cpu(), want)
    small = clouds.uniform(2, 1100, seed=12)                     # fewer points than point_all: point_all = N
    sub2 = np.random.RandomState(1).choice(1100, 1024, False)
    assert torch.equal(EF.sample_points(small.cuda(), 1024, subset=sub2).cpu(), FR.sample_points(small, 1024, sub2))
    with pytest.raises(NotImplementedError):
        EF.sample_points(pts.cuda(), 1000)


def test_step_against_oracle(M):
    """One whole fine-tune iteration (FPS subset -> augment -> forward -> CE -> backward -> clip -> AdamW with layer-wise
    lr decay) on a fresh batch: product on the GPU vs oracle on the CPU, DropPath / Dropout off on both sides."""
    from gm3d_amd import engine_finetune as EF
    torch.manual_seed(0)
    om = R.det_fill_(FR.PointTransformer(drop_path_rate=0.0), seed=5)
    for mod in om.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    pm = build(seed=5, drop_path=0.0)
    om.train(); pm.train()
    lr, args = 5e-4, SimpleNamespace(lr=5e-4, min_lr=1e-6, warmup_epochs=10, epochs=300)
    oopt = torch.optim.AdamW(FR.param_groups_lrd(om), lr=lr)
    popt = EF.build_optimizer(pm, lr=lr)
    for opt in (oopt, popt):
        EF.adjust_learning_rate(opt, 37.25, args)
    B = 6
    pts = clouds.gaussian(B, 2048, seed=21)
    targets = torch.tensor([1, 5, 39, 0, 7, 7])
    subset = np.random.RandomState(3).choice(1200, 1024, False)
    g = torch.Generator().manual_seed(8)
    scale = torch.rand(B, 3, generator=g) * (1.5 - 2.0 / 3.0) + 2.0 / 3.0
    shift = torch.rand(B, 3, generator=g) * 0.4 - 0.2
    pre = {k: v.detach().cpu().clone() for k, v in pm.named_parameters()}
    oloss, oout, ognorm = FR.finetune_step(om, oopt, pts.clone(), targets, 1024, subset, scale, shift, max_norm=10.0)
    # product: keep the gradients for the comparison (update=False), then apply the update separately
    res = EF.finetune_step(pm, nn.CrossEntropyLoss(), popt, pts.clone().cuda(), targets.cuda(), npoints=1024, max_norm=None,
                           bf16=False, subset=subset, aug_draws=(scale, shift), update=False)
    assert rel(res["outputs"], oout) <= 1e-5
    assert rel(res["loss"], oloss) <= 1e-5
    gnorm = torch.nn.utils.clip_grad_norm_(pm.parameters(), 10.0)
    assert rel(gnorm, ognorm) <= 2e-5
    og = dict(om.named_parameters())
    for k, p in pm.named_parameters():
        assert float((p.grad.double().cpu() - og[k].grad.double()).abs().max()) <= 2e-4 * float(og[k].grad.abs().max()) + 1e-5 * float(ognorm), k
    # optimizer arithmetic: the product's own gradients through torch's AdamW on the CPU with the reference's groups
    cpu = {k: torch.nn.Parameter(v.clone()) for k, v in pre.items()}
    for k, p in pm.named_parameters():
        cpu[k].grad = p.grad.detach().cpu().clone()
    groups = {}
    for k, p in cpu.items():
        lid = FR.layer_id(k)
        groups.setdefault((lid, p.ndim == 1), {"params": [], "lr": None, "weight_decay": 0.0 if p.ndim == 1 else 0.05,
                                               "lr_scale": 0.75 ** (12 - lid)})["params"].append(p)
    copt = torch.optim.AdamW(list(groups.values()), lr=lr)
    EF.adjust_learning_rate(copt, 37.25, args)
    copt.step()
    popt.step()
    psd = dict(pm.named_parameters())
    assert max(rel(psd[k], cpu[k]) for k in cpu) <= 1e-6


def test_flat_optimizer_equals_layer_decay_adamw(M):
    """build_optimizer(flat=True) -- one pass over flat buffers with a per-element lr multiplier -- against torch AdamW over
    the reference's layer-decay groups + clip_grad_norm_, three steps with fresh gradients and a moving learning rate."""
    from gm3d_amd import engine_finetune as EF
    args = SimpleNamespace(lr=5e-4, min_lr=1e-6, warmup_epochs=10, epochs=300)
    a, b = build(seed=11), build(seed=11)
    oa = EF.build_optimizer(a, lr=5e-4)
    ob = EF.build_optimizer(b, lr=5e-4, flat=True, max_norm=10.0)
    assert ob.LS is not None and ob.n_decay < ob.n
    nb = dict(b.named_parameters())
    assert float(ob.LS.min()) == pytest.approx(0.75 ** 12) and float(ob.LS.max()) == 1.0
    g = torch.Generator(device="cuda").manual_seed(3)
    for it in range(3):
        for opt in (oa, ob):
            EF.adjust_learning_rate(opt, 3.0 + 9 * it, args)
        for k, p in a.named_parameters():
            gr = torch.randn(p.shape, device="cuda", generator=g) * (3.0 if it == 1 else 0.01)   # step 1 clips, 0 and 2 do not
            p.grad = gr.clone()
            nb[k].grad = gr.clone()
        na = torch.nn.utils.clip_grad_norm_(list(a.parameters()), 10.0)
        oa.step()
        nbn = ob.step()
        assert rel(nbn, na) <= 1e-6
        assert max(rel(nb[k], p) for k, p in a.named_parameters()) <= 1e-6
    from gm3d_amd.fused import weight_cache
    w = nb["blocks.blocks.3.attn.qkv.weight"]
    assert torch.equal(weight_cache.get(w, torch.bfloat16), w.detach().to(torch.bfloat16))


def test_bf16_step_trains(M):
    """Throughput precision: a few bf16 iterations reduce the loss on a fixed batch and stay close to the fp32 logits."""
    from gm3d_amd import engine_finetune as EF
    pm = build(seed=6, drop_path=0.0)
    pm.train()
    pts = clouds.gaussian(8, 2048, seed=31).cuda()
    targets = torch.arange(8).cuda() % 40
    subset = np.random.RandomState(4).choice(1200, 1024, False)
    with torch.no_grad():
        x = EF.sample_points(pts, 1024, subset=subset)
        pm.eval()
        l32 = pm(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            l16 = pm(x).float()
        pm.train()
    assert rel(l16, l32) <= 5e-2
    opt = EF.build_optimizer(pm, lr=1e-3)
    for g in opt.param_groups:
        g["lr"] = 1e-3 * g["lr_scale"]
    losses = []
    for _ in range(6):
        out = EF.finetune_step(pm, nn.CrossEntropyLoss(), opt, pts, targets, npoints=1024, max_norm=10.0, bf16=True,
                               subset=subset, augment=False)
        losses.append(float(out["loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_evaluate_and_checkpoint_roundtrip(tmp_path, M):
    from gm3d_amd import checkpoint as C
    from gm3d_amd import engine_finetune as EF
    pm = build(seed=7)
    pts = clouds.gaussian(5, 2048, seed=41)
    labels = torch.tensor([[1], [2], [3], [4], [5]])
    loader = [(None, None, (pts[:3], labels[:3])), (None, None, (pts[3:], labels[3:]))]
    a = EF.evaluate(loader, pm, "cuda", npoints=1024, bf16=False)
    assert a["n"] == 5 and 0.0 <= a["acc1"] <= 100.0
    path = str(tmp_path / "ft.pth")
    opt = EF.build_optimizer(pm, lr=1e-3)
    C.save_checkpoint(path, pm, opt, epoch=3, model_name="PointTransformer")
    pm2 = build(seed=8)
    opt2 = EF.build_optimizer(pm2, lr=1e-3)
    assert C.load_checkpoint(path, pm2, opt2) == 4
    b = EF.evaluate(loader, pm2, "cuda", npoints=1024, bf16=False)
    assert a == b


def test_graph_replay_equals_eager(M):
    """hipGraph replay of the fine-tune iteration against the same iterations launched eagerly (DropPath / Dropout /
    augmentation off so both sides are deterministic; the FPS subset comes from identically seeded host generators)."""
    from gm3d_amd import engine_finetune as EF
    crit = nn.CrossEntropyLoss()
    batches = [clouds.gaussian(8, 2048, seed=50 + i).cuda() for i in range(3)]
    targets = (torch.arange(8).cuda() * 3) % 40
    args = SimpleNamespace(lr=1e-3, min_lr=1e-6, warmup_epochs=0, epochs=300)

    def run(graphed, flat=False, overlap=False):
        pm = build(seed=9, drop_path=0.0)
        pm.train()
        opt = EF.build_optimizer(pm, lr=1e-3, capturable=True, flat=flat, max_norm=10.0)
        EF.adjust_learning_rate(opt, 5.0, args)
        rng = np.random.RandomState(7)
        losses = []
        if graphed:
            # warm-up iterations are real steps: give the eager side the same three
            g = EF.GraphedFinetuneStep(pm, crit, opt, batches[0], targets, npoints=1024, max_norm=10.0, bf16=False, rng=rng,
                                       augment=False, warmup_iters=3, overlap_sampling=overlap)
            nxt = {id(a): b for a, b in zip(batches, batches[1:])}
            step = (lambda x: g(x, targets, next_points=nxt.get(id(x)))) if overlap else (lambda x: g(x, targets))
        else:
            sub = rng.choice(1200, 1024, False)
            for _ in range(3):          # the 3 warm-up iterations (capture itself executes nothing), on batches[0] with the first subset
                EF.finetune_step(pm, crit, opt, batches[0], targets, npoints=1024, max_norm=10.0, bf16=False, subset=sub,
                                 augment=False)
            step = lambda x: EF.finetune_step(pm, crit, opt, x, targets, npoints=1024, max_norm=10.0, bf16=False,
                                              subset=rng.choice(1200, 1024, False), augment=False)
        # what the allocator hands out now: sentinels (a replay must not write there) and NaN blocks (nor read there)
        guards = [torch.full((n,), 12345.0, device="cuda") for n in (1, 2, 8, 64, 1024, 1 << 18) for _ in range(64)] if graphed else []
        ints = [torch.full((), 777, dtype=torch.int64, device="cuda") for _ in range(128)] if graphed else []
        poison = [torch.full((n,), float("nan"), device="cuda") for n in (1, 4, 16, 96, 384, 1536, 1 << 14, 1 << 20) for _ in range(64)] if graphed else []
        for x in batches:
            losses.append(float(step(x)["loss"]))
        assert all(bool((t == 12345.0).all()) for t in guards) and all(int(t) == 777 for t in ints)
        del poison
        return losses, {k: v.detach().clone() for k, v in pm.named_parameters()}

    le, pe = run(False)
    lg, pg = run(True)
    assert max(abs(a - b) for a, b in zip(le, lg)) <= 1e-5 * max(le)
    assert max(rel(pg[k], pe[k]) for k in pe) <= 1e-5
    # the flat optimizer inside the captured step: replay == eager with the same optimizer; against the torch-AdamW run only
    # loosely (lr 1e-3 on this batch is a violent trajectory -- the loss triples -- that amplifies last-bit differences of the
    # update arithmetic; test_flat_optimizer_equals_layer_decay_adamw pins the arithmetic itself at 1e-6)
    lfe, pfe = run(False, flat=True)
    lf, pf = run(True, flat=True)
    assert lfe == lf, (lfe, lf)                                   # exact: deterministic kernels, same operands
    assert all(torch.equal(pf[k], pfe[k]) for k in pfe)
    assert max(abs(a - b) for a, b in zip(le, lf)) <= 5e-3 * max(le)
    # the sampling graph of batch i+1 replayed on a second stream beside the training graph of batch i: same numbers
    lo, po = run(True, flat=True, overlap=True)
    assert lf == lo, (lf, lo)
    assert all(torch.equal(po[k], pf[k]) for k in pf)
