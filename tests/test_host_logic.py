"""CPU: host-side logic of the product package that needs no kernel launch (mask generation, ranking
loss, schedule, optimizer grouping, EMA, augmentation, sharding), checked against the oracle restatement
and the reference-made fixtures."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import model_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def M():
    from gm3d_amd import models_mae_learn_loss as M
    return M


@pytest.fixture(scope="module")
def model(M):
    torch.manual_seed(0)
    return M.mae_vit_base_patch16_dec512d8b()


def test_state_dict_matches_live_reference_keys(model):
    import json
    man = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    assert set(k for k, _ in model.named_parameters()) == set(man["live_parameters"])
    for k, v in model.state_dict().items():
        assert list(v.shape) == man["state_dict"][k], k
    assert sum(p.numel() for p in model.parameters()) == man["n_live_parameters"] == 36840288
    # a full reference checkpoint (485 keys, DDP 'module.' prefix) loads; the dead keys are reported
    fake = {"module." + k: torch.zeros(s) for k, s in man["state_dict"].items()}
    ignored = model.__class__().load_reference_state_dict(fake)
    assert len(ignored) == 485 - len(model.state_dict())


@pytest.mark.parametrize("case", ["b2_uniform", "b4_gaussian"])
def test_generate_mask_matches_reference(model, case):
    fx = np.load(os.path.join(GOLD, "pretrain_%s.npz" % case))
    lp = torch.from_numpy(fx["teacher_loss_pred"])
    B = lp.shape[0]
    m0 = model.generate_mask(lp, 0.6, epoch=0, total_epoch=400, noise=torch.from_numpy(fx["mask_e0_noise"]))
    assert np.array_equal(m0.numpy(), fx["mask_e0"])
    rng = np.random.RandomState(int(fx["mask_e200_np_seed"]))
    noise = torch.zeros(B, 64)
    order = torch.argsort(lp, dim=1)
    for i in range(B):
        rest = np.delete(np.arange(64), order[i, -9:].numpy())
        rng.shuffle(rest)
        noise[i, torch.from_numpy(rest)] = torch.arange(len(rest), dtype=torch.float32)
    m200 = model.generate_mask(lp, 0.6, epoch=200, total_epoch=400, noise=noise)
    assert np.array_equal(m200.numpy(), fx["mask_e200"])


def test_generate_mask_properties(model):
    torch.manual_seed(1)
    lp = torch.randn(16, 64)
    for epoch, forced in ((0, 0), (199, 9), (399, 19)):
        m = model.generate_mask(lp, 0.6, epoch=epoch, total_epoch=400)
        assert m.shape == (16, 64) and set(m.unique().tolist()) <= {0.0, 1.0}
        assert (m.sum(1) == 39).all()
        if forced:
            top = torch.argsort(lp, dim=1)[:, -forced:]
            assert (torch.gather(m, 1, top) == 1).all()     # hardest patches are always masked
    a = model.generate_mask(lp, 0.6, epoch=0, total_epoch=400)
    b = model.generate_mask(lp, 0.6, epoch=0, total_epoch=400)
    assert not torch.equal(a, b)                             # fresh randomness per call


def test_learning_loss_matches_oracle(model):
    o = R.PointMAEGM3D.forward_learning_loss
    torch.manual_seed(2)
    p, t = torch.randn(5, 39), torch.rand(5, 39)
    t[:, 3] = t[:, 4]                                        # ties contribute nothing
    for relative in (True, False):
        assert torch.allclose(model.forward_learning_loss(p, None, t, relative=relative), o(None, p, None, t, relative),
                              rtol=1e-6, atol=0)
    assert torch.isnan(model.forward_learning_loss(p, None, torch.ones(5, 39), relative=True))  # 0/0 like the reference


def test_split_and_take(M):
    mask = torch.tensor([[True, False, False, True, False], [False, True, True, False, False]])
    vis, msk = M.split_ids(mask, 3)
    assert vis.tolist() == [[1, 2, 4], [0, 3, 4]] and msk.tolist() == [[0, 3], [1, 2]]
    x = torch.arange(2 * 5 * 2, dtype=torch.float32).view(2, 5, 2)
    assert torch.equal(M.take(x, vis), x[~mask].reshape(2, -1, 2))       # == boolean-mask indexing order
    assert torch.equal(M.take(x, msk), x[mask].reshape(2, -1, 2))
    assert M.split_ids(mask)[0].shape == (2, 3)


def test_lr_schedule_optimizer_groups_and_ema(model):
    from gm3d_amd import engine_pretrain as E
    fx = np.load(os.path.join(GOLD, "lr_sched.npz"))
    args = SimpleNamespace(lr=1e-3, min_lr=0.0, warmup_epochs=40, epochs=400)
    opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, fused=False)
    got = [E.adjust_learning_rate(opt, float(e), args) for e in fx["epochs"]]
    assert np.allclose(got, fx["lrs"], rtol=1e-12, atol=0)
    assert all(g["lr"] == got[-1] for g in opt.param_groups)
    no_decay, decay = opt.param_groups
    assert no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == 0.05
    names = {id(p): n for n, p in model.named_parameters()}
    nd = {names[id(p)] for p in no_decay["params"]}
    assert "mask_token" in nd and "norm_p.weight" in nd and "blocks.blocks.0.mlp.fc1.bias" in nd
    assert all(names[id(p)].endswith("weight") and p.dim() > 1 for p in decay["params"])
    ref_groups = R.param_groups(model, 0.05)
    assert [len(g["params"]) for g in ref_groups] == [len(no_decay["params"]), len(decay["params"])]
    assert abs(E.ema_decay_for_epoch(50) - R.ema_decay_for_epoch(50)) < 1e-15 and E.ema_decay_for_epoch(150) == 0.9999

    small = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    e1, e2 = E.ModelEma(small, decay=0.9), R.ModelEma(small, decay=0.9)
    with torch.no_grad():
        for p in small.parameters():
            p.add_(1.0)
        small[1].num_batches_tracked.add_(7)
    e1.update(small); e2.update(small)
    for (k, a), (_, b) in zip(e1.ema.state_dict().items(), e2.ema.state_dict().items()):
        assert torch.allclose(a.float(), b.float(), rtol=1e-6), k
    assert not e1.ema.training and not any(p.requires_grad for p in e1.ema.parameters())
    e3 = E.ModelEma(small, decay=0.5)
    e3.update(torch.nn.parallel.DataParallel(small) if False else SimpleNamespace(state_dict=lambda: {"module." + k: v for k, v in small.state_dict().items()}))


def test_scale_and_translate():
    from gm3d_amd import engine_pretrain as E
    fx = np.load(os.path.join(GOLD, "pretrain_b2_uniform.npz"))
    pc = torch.from_numpy(fx["pts"]).clone()
    out = E.train_transforms(pc, draws=(torch.from_numpy(fx["scale"]), torch.from_numpy(fx["shift"])))
    assert out is pc and np.allclose(out.numpy(), fx["samples"], rtol=1e-7, atol=1e-7)   # in place, like the reference
    pc = torch.zeros(512, 8, 3) + 1.0
    E.train_transforms(pc)
    v = pc[:, 0]            # = scale + shift per cloud; scale in [2/3,3/2], shift in [-0.2,0.2]
    assert (v >= 2 / 3 - 0.2 - 1e-6).all() and (v <= 1.5 + 0.2 + 1e-6).all() and v.std() > 0.1
    assert torch.equal(pc[:, 0], pc[:, 7])


def test_shard_for_rank():
    from gm3d_amd.engine_pretrain import shard_for_rank
    parts = [shard_for_rank(103, r, 4, epoch=3, seed=1) for r in range(4)]
    assert all(len(p) == 26 for p in parts)
    assert set(torch.cat(parts).tolist()) == set(range(103))
    assert not torch.equal(shard_for_rank(103, 0, 4, epoch=4, seed=1), parts[0])      # set_epoch reshuffles


def test_collection_from_the_repo_root_needs_no_gpu():
    """`python -m pytest --collect-only` from the repo ROOT (no `tests` argument) must succeed on a machine without a GPU:
    pytest.ini restricts collection to tests/ (a diagnostic under tools/ once matched *_test.py and made HIP calls at import,
    which also broke the fresh-parent invariant of tests/conftest.py::pytest_collection_finish on the GPU box), and no test
    module may touch the GPU at import (conftest asserts that before it starts the data-parallel workers)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")      # a GPU box looks like a CPU box to the child
    r = subprocess.run([sys.executable, "-m", "pytest", "--collect-only", "-q", "-p", "no:cacheprovider"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "tools/" not in r.stdout and "error" not in r.stdout.lower().split("\n")[-2]


def test_gemm_table_names_a_hand_written_kernel_for_every_shape_of_the_path():
    """gemm.choose (the per-shape kernel table read off profiles/r03_gemm_kbench.txt) never answers "lib": every plain product of the
    bf16 step -- block stacks at the student's / teacher's row counts, mini-PointNet, heads -- has a hand-written kernel, and the answer
    is one mm() knows how to dispatch."""
    from gm3d_amd import gemm
    shapes = [(M, N, K) for M in (3200, 3328, 4096, 8192) for (N, K) in ((1152, 384), (384, 384), (1536, 384), (384, 1536), (384, 1152))]
    shapes += [(262144, 256, 128), (262144, 512, 256), (262144, 384, 512), (102400, 384, 512), (102400, 512, 384), (262144, 256, 512),
               (262144, 128, 256), (8192, 512, 256), (8192, 256, 512), (8192, 384, 128), (8192, 128, 384), (8192, 1024, 384), (8192, 384, 1024)]
    for M, N, K in shapes:
        how = gemm.choose(M, N, K)
        assert how != "lib" and (how == "own" or how.startswith("ring") or how.startswith("dma")), (M, N, K, how)
        if how.startswith("dma"):
            bm, bn = how[3:].split("x")
            assert int(bm) in (64, 128) and N % int(bn) == 0, (M, N, K, how)


def test_drop_path_scales_are_drawn_per_site_and_consumed_in_plan_order(M):
    """prepare_drop_path draws the factors of several stacks in one go and drop_path_scales hands them out in that order; a request
    that does not match the plan falls back to a fresh draw (host logic only: CPU tensors)."""
    import torch
    dev = torch.device("cpu")
    torch.manual_seed(0)
    plan = [[0.0, 0.0, 0.1, 0.1], [0.05, 0.05]]
    M.prepare_drop_path(4, plan, True, dev)
    a = M.drop_path_scales(4, plan[0], True, dev)
    b = M.drop_path_scales(4, plan[1], True, dev)
    assert a[0] is None and a[1] is None and a[2].shape == (4,) and b[0].shape == (4,)
    for t, p in ((a[2], 0.1), (a[3], 0.1), (b[0], 0.05), (b[1], 0.05)):          # floor(keep + u) / keep is 0 or 1 / keep
        keep = 1.0 - p
        assert bool(((t == 0) | ((t - 1.0 / keep).abs() < 1e-6)).all())
    c = M.drop_path_scales(4, [0.2], True, dev)                                   # no plan left: drawn on the spot
    assert c[0].shape == (4,)
    assert M.drop_path_scales(4, [0.2], False, dev) == [None]                      # eval mode: inactive


def test_run_epoch_without_a_model_key():
    """run_epoch's warm-up counter is a weak dictionary; callers that omit model_key must not hit a TypeError."""
    from gm3d_amd import engine_pretrain as E
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([w], lr=0.1)
    args = SimpleNamespace(accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=1, epochs=4)
    seen = []

    def eager_step(samples, it):
        seen.append(it)
        z = samples.sum() * 0
        return {"loss": z + 1.0, "loss_learn": z + 0.5, "loss_mse": z, "loss_chfr": z + 2.0, "grad_norm": z + 3.0}

    loader = [torch.ones(2, 8, 3) for _ in range(4)]
    stats = E.run_epoch(loader, opt, torch.device("cpu"), 0, args, eager_step, capture=lambda ex: None, print_freq=100)
    assert seen == [0, 1, 2, 3]
    assert abs(stats["loss"] - 1.5) < 1e-6 and abs(stats["grad_norm"] - 3.0) < 1e-6 and stats["replayed_iters"] == 0


def test_roofline_bookkeeping_has_both_bounds_and_the_named_kernels():
    """gm3d_gemm_nt_bf16_multi, gm3d_patch_chamfer_loss_fwd/bwd and gm3d_adamw_ema_flat_step have algorithmic
    figures, and a GEMM-class kernel is graded against the roof that gives the larger time (flops vs operand bytes)."""
    import bench
    # the stacks' weight-gradient launch of the north-star step: 12 products over 8192 / 3328 rows
    probs = [(12, 8192, 1536, 384), (12, 8192, 384, 1536), (12, 8192, 384, 384), (12, 8192, 1152, 384)]
    meta = {"count": 4, "problems": probs}
    both = bench.algorithmic_both("gm3d_gemm_nt_bf16_multi", meta)
    flop = sum(2.0 * b * r * n * k for b, r, n, k in probs)
    assert both["flop"] == flop and both["bytes"] == sum(b * (2.0 * r * (n + k) + 4.0 * n * k) for b, r, n, k in probs)
    assert both["bound"] == ("mfma" if flop / 2.5e15 >= both["bytes"] / 8e12 else "hbm")
    # a skinny product (65,536 x 96 -> 288) is an HBM stream, a square one is matrix-core work
    assert bench.algorithmic("gm3d_gemm_tn_bf16_ws", {"M": 65536, "N": 288, "K": 96})[0] == "hbm"
    assert bench.algorithmic("gm3d_gemm_tn_bf16_dmaw", {"M": 8192, "N": 1536, "K": 1536})[0] == "mfma"
    # (fc1 of the north-star blocks, 8192 x 384 -> 1536, sits just below the ridge of 312 FLOP/B: 296 -- the operand bytes bind)
    assert bench.algorithmic("gm3d_gemm_tn_bf16_dmaw", {"M": 8192, "N": 1536, "K": 384})[0] == "hbm"
    for name, m in (("gm3d_patch_chamfer_loss_fwd", {"B": 128, "M": 39, "dtype": "torch.float32"}),
                    ("gm3d_patch_chamfer_loss_bwd", {"B": 128, "M": 39, "dtype": "torch.float32"}),
                    ("gm3d_adamw_ema_flat_step", {"n": 36_000_000, "ema": True})):
        b, amount, unit = bench.algorithmic(name, m)
        assert b == "hbm" and unit == "B" and amount > 0
    # mixed launches of one kernel: work summed per roof, one bound for the whole
    per = [(0.010, {"M": 65536, "N": 288, "K": 96}), (0.020, {"M": 8192, "N": 384, "K": 1536})]
    r = bench.roofline_of("gm3d_gemm_tn_bf16_ring", per, 0.030)
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1 and {"tflops", "gbs", "frac_mfma", "frac_hbm"} <= set(r)
    assert abs(r["frac"] - max(r["frac_mfma"], r["frac_hbm"])) < 1e-3
