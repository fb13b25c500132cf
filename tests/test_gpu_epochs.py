"""-m gpu: the three `train_one_epoch` drop-ins run end to end on tiny synthetic loaders with the reference's call signatures
(P/engine_pretrain.py:25-31, P/engine_finetune.py:70-74, P/engine_pretrain_Classifier_SVM.py:40-45): finite stats, learning-rate
schedule applied, parameters move, EMA follows."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

from tests import clouds

pytestmark = pytest.mark.gpu


class Loader(list):
    pass


def test_pretrain_train_one_epoch():
    from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
    torch.manual_seed(0)
    model = M.mae_vit_base_patch16_dec512d8b().cuda()
    ema = E.ModelEma(model, 0.999)
    opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40,
                           learning_loss=True)
    loader = Loader(clouds.gaussian(8, 1024, seed=i) for i in range(3))
    before = model.blocks.blocks[0].attn.qkv.weight.detach().clone()
    ema_before = ema.ema.blocks.blocks[0].attn.qkv.weight.detach().clone()
    stats = E.train_one_epoch(model, loader, opt, torch.device("cuda"), 50, None, log_writer=None, args=args, model_ema=ema)
    assert set(stats) >= {"loss", "loss_learn", "loss_chfr", "grad_norm", "lr"}
    assert all(v == v and abs(v) != float("inf") for v in stats.values())
    assert ema.decay == E.ema_decay_for_epoch(50)
    assert not torch.equal(model.blocks.blocks[0].attn.qkv.weight, before)
    assert not torch.equal(ema.ema.blocks.blocks[0].attn.qkv.weight, ema_before)


def _pretrain_setup(B, accum=1, flat=True, bn_eval=False, drop_path=True):
    from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
    torch.manual_seed(0)
    model = M.mae_vit_base_patch16_dec512d8b().cuda()
    if not drop_path:
        for mod in model.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
    ema = E.ModelEma(model, 0.999)
    opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema) if flat else \
        E.build_optimizer(model, lr=1e-3, weight_decay=0.05)
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=accum, lr=1e-3, min_lr=0.0,
                           warmup_epochs=40, learning_loss=True)
    return E, model, ema, opt, args


def test_train_one_epoch_runs_the_measured_configuration():
    """train_one_epoch at B=128 bf16 must run what bench.py measures: hipGraph replays of the whole step.  Its steady-state rate
    (the replayed iterations, capture excluded and reported separately) within 10 % of replaying GraphedPretrainStep directly."""
    import time
    B, n = 128, 40
    E, model, ema, opt, args = _pretrain_setup(B)
    loader = Loader(clouds.gaussian(B, 1024, seed=i).cuda() for i in range(4)) * (n // 4)
    E._warm.clear()
    s0 = E.train_one_epoch(model, loader, opt, torch.device("cuda"), 200, None, args=args, model_ema=ema, print_freq=20)
    assert s0["replayed_iters"] == n - E.EAGER_WARMUP_ITERS          # first epoch: the first iterations are the eager warm-up
    s1 = E.train_one_epoch(model, loader, opt, torch.device("cuda"), 201, None, args=args, model_ema=ema, print_freq=20)
    assert s1["replayed_iters"] == n and s1["capture_s"] > 0         # later epochs: captured before the first iteration
    assert all(v == v and abs(v) != float("inf") for v in s1.values())
    # the bench figure: the same step object replayed directly (bench.py's timed loop)
    g = E.GraphedPretrainStep(model, ema, opt, args, loader[0], 201, warmup_iters=0)
    for i in range(5):
        g(loader[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        E.adjust_learning_rate(opt, 201 + i / n, args)
        g(loader[i % 4])
    torch.cuda.synchronize()
    bench = B * n / (time.perf_counter() - t0)
    print("epoch replay %.0f clouds/s (whole epoch incl. capture %.0f, capture %.2f s), direct replay %.0f clouds/s"
          % (s1["replay_clouds_per_s"], s1["clouds_per_s"], s1["capture_s"], bench))
    assert s1["replay_clouds_per_s"] >= 0.9 * bench, (s1, bench)


def test_accum_iter_two_equals_reference_loop_and_concatenated_batch():
    """args.accum_iter = 2 (P/engine_pretrain.py:72-73,195-212 + P/util/misc.py:256-270: lr set, and clip / AdamW / EMA / zero_grad
    run, once per window of 2 iterations; gradients of loss/2 summed).
    (a) the flat optimizer's window (eager AND hipGraph replay) == the reference's loop written out with torch.optim.AdamW and
        p.grad accumulation; (b) one window on two half-batches == ONE step on the concatenated batch -- with DropPath off and
        the student's BatchNorm layers in eval mode: train-mode BatchNorm normalises each micro-batch with its own statistics, in
        the reference as here, so with it the two are different computations."""
    from gm3d_amd import engine_pretrain as E
    B = 8
    data = [clouds.gaussian(B, 1024, seed=40 + i).cuda() for i in range(4)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(70 + i)).cuda() for i in range(4)]

    def reference_loop():
        E_, model, ema, opt, args = _pretrain_setup(B, accum=2, flat=False, drop_path=False)
        model.train(True)
        gn, g1 = [], None
        for it in range(4):
            if it % 2 == 0:
                E_.adjust_learning_rate(opt, 200 + it / 4, args)
                opt.zero_grad(set_to_none=True)
            out = E_.step_forward_backward(model, ema, data[it].clone(), 200, args, mask_noise=noise[it], augment=False,
                                           optimizer=opt, accum_first=(it % 2 == 0), accum_last=(it % 2 == 1))
            if it % 2 == 1:
                if g1 is None:
                    g1 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
                gn.append(float(E_.step_update(model, ema, opt)))
        return g1, gn

    ref_g1, ref_gn = reference_loop()
    for mode in ("eager", "graph"):
        E_, model, ema, opt, args = _pretrain_setup(B, accum=2, drop_path=False)
        model.train(True)
        gn, g1 = [], None
        slot = {id(p): v for p, v in opt.flat_grad_views()}
        if mode == "graph":
            g = E_.GraphedPretrainStep(model, ema, opt, args, data[0], 200, warmup_iters=0, augment=False, inject_mask_noise=True)
        for it in range(4):
            if it % 2 == 0:
                E_.adjust_learning_rate(opt, 200 + it / 4, args)
            if mode == "graph":
                out = g(data[it], noise[it], update=(it % 2 == 1))
            else:
                out = E_.pretrain_step(model, ema, opt, data[it].clone(), 200, args, mask_noise=noise[it], augment=False, micro_step=it)
            assert (out["grad_norm"] is None) == (it % 2 == 0)
            if it % 2 == 1:
                gn.append(float(out["grad_norm"]))
                if g1 is None:      # the window's summed gradient (the update kernel reads G, it does not rewrite it)
                    g1 = {k: slot[id(p)].detach().clone() for k, p in model.named_parameters()}
        for a, b in zip(gn, ref_gn):
            assert abs(a - b) <= 2e-2 * abs(b), (mode, gn, ref_gn)
        # first window, element by element (bf16 GEMMs on both sides; the fused wgrad GEMMs sum in another order)
        worst = max(float((g1[k] - v).abs().max() / v.abs().max().clamp_min(1e-6)) for k, v in ref_g1.items())
        assert worst <= 3e-2, (mode, worst)

    # (b) window of two half-batches vs one step on the concatenated batch, BatchNorm on running statistics
    from gm3d_amd import models_mae_learn_loss as M
    res = {}
    was = (M.FUSED_EMBED, M.FUSED_HEADS)
    M.FUSED_EMBED = M.FUSED_HEADS = False     # the per-op modules honour BatchNorm1d.eval() under autograd (the fused nodes are train-mode only)
    try:
        for name in ("window", "concat"):
            E_, model, ema, opt, args = _pretrain_setup(B, accum=2 if name == "window" else 1, drop_path=False)
            model.train(True)
            for mod in model.modules():
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.eval()
            E_.adjust_learning_rate(opt, 200.0, args)
            if name == "window":
                for it in range(2):
                    out = E_.pretrain_step(model, ema, opt, data[it].clone(), 200, args, mask_noise=noise[it], augment=False,
                                           micro_step=it)
            else:
                out = E_.pretrain_step(model, ema, opt, torch.cat(data[:2]).clone(), 200, args, mask_noise=torch.cat(noise[:2]),
                                       augment=False)
            res[name] = (float(out["grad_norm"]), opt.P.clone())
    finally:
        M.FUSED_EMBED, M.FUSED_HEADS = was
    assert abs(res["window"][0] - res["concat"][0]) <= 2e-2 * res["concat"][0], res
    assert float((res["window"][1] - res["concat"][1]).abs().max() / res["concat"][1].abs().max()) <= 2e-2


class _SnapshotLoader:
    """Yields the batches; just before the last one it snapshots the model's buffers (stream-ordered clones)."""

    def __init__(self, batches, model):
        self.batches, self.model, self.snap = batches, model, None

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for i, b in enumerate(self.batches):
            if i == len(self.batches) - 1:
                self.snap = [t.detach().clone() for t in self.model.buffers()]
            yield b


def test_partial_accumulation_window_is_discarded_at_the_epoch_boundary():
    """len(data_loader) % accum_iter != 0: the reference calls optimizer.zero_grad() at the start of every epoch
    (P/engine_pretrain.py:62), so the trailing micro-batch of epoch e never reaches an update.  Two runs from the same state --
    a 3-iteration epoch followed by a 2-iteration epoch, differing ONLY in the stray third batch -- must end with identical
    parameters: the stray batch's other trace, the student's BatchNorm running statistics, is rolled back to what it was before
    that batch, so only a leak of its gradient into the next epoch's first window could tell the runs apart (the flat
    optimizer's GA buffer used to carry it).  Eager and hipGraph replay."""
    from gm3d_amd import engine_pretrain as E
    B = 4
    data = [clouds.gaussian(B, 1024, seed=90 + i).cuda() for i in range(6)]
    for use_graph in (False, True):
        ends = []
        for stray in (2, 5):
            E_, model, ema, opt, args = _pretrain_setup(B, accum=2, drop_path=False)
            torch.manual_seed(5)            # mask noise / augmentation draws: same sequence in both runs
            E._warm.clear()
            loader = _SnapshotLoader([data[i].clone() for i in (0, 1, stray)], model)    # (the augmentation works in place)
            E_.train_one_epoch(model, loader, opt, torch.device("cuda"), 10, None, args=args, model_ema=ema, use_graph=False)
            with torch.no_grad():
                for t, v in zip(model.buffers(), loader.snap):
                    t.copy_(v)
            torch.manual_seed(6)
            if use_graph:
                E._warm[model] = 10 * E.EAGER_WARMUP_ITERS      # capture before the first iteration of the second epoch
            s = E_.train_one_epoch(model, Loader([data[3].clone(), data[4].clone()]), opt, torch.device("cuda"), 11, None, args=args,
                                   model_ema=ema, use_graph=use_graph)
            assert s["replayed_iters"] == (2 if use_graph else 0)
            ends.append(opt.P.clone())
        assert torch.equal(ends[0], ends[1]), (use_graph, float((ends[0] - ends[1]).abs().max()))


def test_finetune_train_one_epoch_and_evaluate():
    from gm3d_amd import engine_finetune as EF
    from gm3d_amd.point_transformer import PointTransformer
    torch.manual_seed(0)
    model = PointTransformer(dict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64,
                                  encoder_dims=384)).cuda()
    opt = EF.build_optimizer(model, lr=5e-4)
    args = SimpleNamespace(lr=5e-4, min_lr=1e-6, warmup_epochs=10, epochs=300, accum_iter=1, bf16=True)
    loader = Loader((None, None, (clouds.gaussian(8, 2048, seed=10 + i), torch.arange(8) % 40)) for i in range(3))
    before = model.cls_head_finetune[0].weight.detach().clone()
    stats = EF.train_one_epoch(model, nn.CrossEntropyLoss(), loader, opt, torch.device("cuda"), 20, None, 10.0, None, log_writer=None,
                               args=args, npoints=1024, print_freq=1)
    assert stats["loss"] == stats["loss"] and stats["lr"] > 0
    assert not torch.equal(model.cls_head_finetune[0].weight, before)
    # the same epoch as graph replays with the flat layer-decay optimizer and the sampling of batch i+1 overlapped with batch i;
    # the loader ends on a smaller batch, which runs eagerly
    model2 = PointTransformer(dict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64,
                                   encoder_dims=384)).cuda()
    opt2 = EF.build_optimizer(model2, lr=5e-4, flat=True, max_norm=10.0)
    EF.adjust_learning_rate(opt2, 20.0, args)
    ex = clouds.gaussian(8, 2048, seed=10).cuda()
    g = EF.GraphedFinetuneStep(model2, nn.CrossEntropyLoss(), opt2, ex, (torch.arange(8) % 40).cuda(), npoints=1024, max_norm=10.0,
                               overlap_sampling=True)
    items = [(None, None, (clouds.gaussian(8, 2048, seed=10 + i), torch.arange(8) % 40)) for i in range(3)]
    items.append((None, None, (clouds.gaussian(5, 2048, seed=19), torch.arange(5) % 40)))
    before2 = model2.cls_head_finetune[0].weight.detach().clone()
    stats2 = EF.train_one_epoch(model2, nn.CrossEntropyLoss(), Loader(iter(items)), opt2, torch.device("cuda"), 20, None, 10.0, None,
                                log_writer=None, args=args, npoints=1024, print_freq=1, step=g)
    assert stats2["loss"] == stats2["loss"] and stats2["lr"] > 0
    assert not torch.equal(model2.cls_head_finetune[0].weight, before2)
    ev = EF.evaluate([(None, None, (clouds.gaussian(4, 2048, seed=3), torch.arange(4).view(4, 1)))], model, "cuda", npoints=1024)
    assert ev["n"] == 4 and 0.0 <= ev["acc1"] <= 100.0


def test_published_train_one_epoch():
    from gm3d_amd import engine_pretrain_Classifier_SVM as EV
    from gm3d_amd import models_mae_learn_loss_Classifier_SVM_feature_besed as V
    from gm3d_amd.point_mae import Point_MAE
    torch.manual_seed(0)
    model = V.mae_vit_base_patch16_dec512d8b().cuda()
    teacher = Point_MAE({"group_size": 32, "num_group": 64, "loss": "cdl2",
                         "transformer_config": {"mask_ratio": 0, "mask_type": "rand", "trans_dim": 384, "encoder_dims": 384, "depth": 12,
                                                "drop_path_rate": 0.1, "num_heads": 6, "decoder_depth": 4, "decoder_num_heads": 6}}).cuda()
    for p in teacher.parameters():
        p.requires_grad_(False)
    ema = EV.ModelEma(model, 0.999)
    opt = EV.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
    args = SimpleNamespace(mask_ratio=0.6, epochs=300, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=10,
                           learning_loss=True)
    loader = Loader(clouds.gaussian(8, 1024, seed=20 + i) for i in range(2))
    stats = EV.train_one_epoch(model, None, loader, None, None, opt, None, torch.device("cuda"), 20, None, log_writer=None, args=args,
                               model_ema=ema, model_teacher=teacher, after_200_epoch=False, classification=False,
                               loss_multiply_by=[13.889, 1000], after_epoch=15, shared_learnable_tokens=False)
    assert all(v == v and abs(v) != float("inf") for v in stats.values())
    with pytest.raises(NotImplementedError):
        EV.train_one_epoch(model, None, loader, None, None, opt, None, torch.device("cuda"), 20, args=args, model_ema=ema,
                           model_teacher=teacher, classification=True)
    # Point_MAE's own pre-training loss (random mask path) runs too
    pm = Point_MAE({"group_size": 32, "num_group": 64, "loss": "cdl2",
                    "transformer_config": {"mask_ratio": 0.6, "mask_type": "rand", "trans_dim": 384, "encoder_dims": 384, "depth": 12,
                                           "drop_path_rate": 0.1, "num_heads": 6, "decoder_depth": 4, "decoder_num_heads": 6}}).cuda()
    loss = pm(clouds.gaussian(4, 1024, seed=1).cuda())
    loss.backward()
    assert float(loss) == float(loss) and pm.increase_dim[0].weight.grad is not None
