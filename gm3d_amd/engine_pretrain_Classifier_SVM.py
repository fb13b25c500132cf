"""One iteration of the published GM3D run: Point-MAE_SA3D/engine_pretrain_Classifier_SVM.py:40-332
(train_one_epoch) with forward_features_dino_decoder (:669-687), as wired by main_pretrain.py:299-328,462-478
(`--learn_feature_loss dino`: a frozen pre-trained Point-MAE supplies target features and decodes points).

  augment -> EMA teacher forward (all 64 tokens, 12-block loss-prediction decoder) -> guided mask ->
  student forward (25 visible / 39 masked) -> frozen Point-MAE: features of all tokens, its decoded points, and points
  decoded from the student's masked-token features (all under no_grad, like the reference :207-211) ->
  loss = w_mse * MSE(normalised features) + w_chfr * Chamfer(points)   [w = 1,1 before `after_epoch`, then
  `loss_multiply_by` = 13.889, 1000]  + pairwise ranking loss of the loss predictor -> backward -> clip/AdamW/EMA.

FPS + KNN run once and are shared by the three networks (identical `samples`).  The online-classifier branch
(`classification=True`, off by default in the reference) is not part of this path.
"""
from contextlib import nullcontext

import torch
import torch.distributed as dist

from . import engine_pretrain as E
from .engine_pretrain import (adjust_learning_rate, ema_decay_for_epoch, step_update, train_transforms,  # noqa: F401
                              GraphedPretrainStep, ModelEma, build_optimizer)


def loss_weights(epoch, args):
    """P/:239-242: plain sum before `after_epoch`, `loss_multiply_by` afterwards."""
    if epoch < getattr(args, "after_epoch", 15):
        return 1.0, 1.0
    w = getattr(args, "loss_multiply_by", (13.889, 1000.0))
    return float(w[0]), float(w[1])


def step_forward_backward(model, model_ema, samples, epoch, args, model_teacher=None, grad_sync=None, mask_noise=None,
                          augment=True, aug_draws=None, optimizer=None, accum_first=True, accum_last=True):
    assert model_teacher is not None, "the published run needs the frozen Point-MAE teacher (--learn_feature_loss dino)"
    raw = model.module if hasattr(model, "module") else model
    teacher = model_ema.ema
    L = raw.num_group
    len_keep = int(L * (1 - args.mask_ratio))
    shared = bool(getattr(args, "shared_learnable_tokens", False))
    if augment:
        samples = train_transforms(samples, draws=aug_draws)
    bf16 = getattr(args, "bf16", False)
    if bf16:
        from .fused import weight_cache
        weight_cache.pin(raw)
        weight_cache.pin(teacher)
        weight_cache.pin(model_teacher, static=True)     # frozen: cast once
        weight_cache.refresh()
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else nullcontext()
    B = samples.shape[0]
    visible_mask = torch.zeros(B, L, dtype=torch.bool, device=samples.device)
    with amp:
        with torch.no_grad():
            group = teacher.group_divider(samples)
            all_ids = (E._arange_ids(B, L, samples.device), E._arange_ids(B, 0, samples.device))
            outs_ema = teacher(samples, mask=visible_mask, shared_learnable_tokens=shared, group=group, ids=all_ids)
            mask, vis_ids, mask_ids = teacher.generate_mask_ids(
                outs_ema["loss_pred"], mask_ratio=args.mask_ratio, guide=True, epoch=epoch, total_epoch=args.epochs,
                after_200_epoch=getattr(args, "after_200_epoch", False), noise=mask_noise)
            bool_masked_pos = mask.flatten(1).to(torch.bool)
        ids = (vis_ids, mask_ids)
        outs = model(samples, mask=bool_masked_pos, shared_learnable_tokens=shared, num_visible=len_keep, group=group, ids=ids)
        M = outs["mask_num"]
        feat = outs["pix_pred"][:, -M:]
        with torch.no_grad():
            feature_target, point_target, point_reconstructed = model_teacher.features_decoder(
                group[0], group[1], feat.detach(), mask_ids)
        loss_outs = raw.forward_loss(feat, feature_target.detach(), outs["mask"], point_target, point_reconstructed,
                                     mask_ids=mask_ids)
        loss_mse, loss_chfr = loss_outs["MSE_mean"], loss_outs["Chamfer_mean"]
        w_mse, w_chfr = loss_weights(epoch, args)
        loss = w_mse * loss_mse + w_chfr * loss_chfr
        loss_learn = raw.forward_learning_loss(outs["loss_pred"][:, -M:], bool_masked_pos, loss_outs["matrix"].detach(),
                                               relative=args.relative)
    accum = getattr(args, "accum_iter", 1)
    total = loss + loss_learn if accum == 1 else (loss + loss_learn) / accum
    E.backward_and_collect(total, raw, optimizer, grad_sync, accum, accum_first, accum_last, async_w=False)
    return {"loss": loss.detach(), "loss_learn": loss_learn.detach(), "loss_chfr": loss_chfr.detach(),
            "loss_mse": loss_mse.detach(), "mask": bool_masked_pos, "matrix": loss_outs["matrix"].detach(),
            "teacher_loss_pred": outs_ema["loss_pred"]}


def pretrain_step(model, model_ema, model_teacher, optimizer, samples, epoch, args, grad_sync=None, mask_noise=None,
                  augment=True, aug_draws=None, clip_grad=5.0, micro_step=None):
    """One iteration; args.accum_iter > 1 needs micro_step (see engine_pretrain.pretrain_step: same window rule, P/:300-316)."""
    accum = getattr(args, "accum_iter", 1)
    if accum > 1 and micro_step is None:
        raise ValueError("args.accum_iter = %d needs micro_step (the iteration index) to place the update" % accum)
    first = accum == 1 or micro_step % accum == 0
    last = accum == 1 or (micro_step + 1) % accum == 0
    if grad_sync is not None:
        overlap_was = grad_sync.overlap
        E.broadcast_buffers(model.module if hasattr(model, "module") else model, group=grad_sync.group)
    out = step_forward_backward(model, model_ema, samples, epoch, args, model_teacher=model_teacher, grad_sync=grad_sync,
                                mask_noise=mask_noise, augment=augment, aug_draws=aug_draws, optimizer=optimizer,
                                accum_first=first, accum_last=last)
    if grad_sync is not None:
        grad_sync.overlap = overlap_was
    if not last:
        out["grad_norm"] = None
        return out
    if grad_sync is not None:
        grad_sync.finish()
    out["grad_norm"] = step_update(model, model_ema, optimizer, clip_grad)
    return out


def graphed_step(model, model_ema, model_teacher, optimizer, args, example, epoch, **kw):
    """hipGraph replay of the whole iteration (see engine_pretrain.GraphedPretrainStep)."""
    return GraphedPretrainStep(model, model_ema, optimizer, args, example, epoch, fwd_bwd=step_forward_backward,
                               extra={"model_teacher": model_teacher}, **kw)


def train_one_epoch(model, classifier, data_loader, data_loader_classifier, criterion_cls, optimizer, optimizer_cls, device,
                    epoch, loss_scaler=None, log_writer=None, args=None, model_ema=None, model_teacher=None, scheduler=None,
                    optimizer_learn_loss=None, after_200_epoch=None, classification=None, loss_multiply_by=None,
                    after_epoch=None, shared_learnable_tokens=None, grad_sync=None, print_freq=20, use_graph=None):
    """Reference signature (P/:40-45).  Averaged stats like P/:330-332.  Iterations run as hipGraph replays under the same
    conditions as engine_pretrain.train_one_epoch (flat optimizer, GPU), captured once per epoch."""
    if classification:
        raise NotImplementedError("the online classifier branch (classification=True) is outside this path")
    assert args.learning_loss and model_ema is not None and model_teacher is not None
    for k, v in (("after_200_epoch", after_200_epoch), ("loss_multiply_by", loss_multiply_by), ("after_epoch", after_epoch),
                 ("shared_learnable_tokens", shared_learnable_tokens)):
        if v is not None:
            setattr(args, k, v)
    import os
    model.train(True)
    model_teacher.eval()
    model_ema.decay = ema_decay_for_epoch(epoch)          # P/:61-66, same schedule as the north-star engine
    device = torch.device(device)
    flat = hasattr(optimizer, "flat_grad_views")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1 and grad_sync is None and flat:
        grad_sync = getattr(optimizer, "_grad_sync", None)
        if grad_sync is None:
            grad_sync = optimizer._grad_sync = E.GradSync.from_flat(optimizer, bucket_bytes=256 << 20)
    if use_graph is None:
        use_graph = flat and device.type == "cuda" and os.environ.get("GM3D_EAGER_EPOCH") != "1"

    def eager_step(samples, it):
        return pretrain_step(model, model_ema, model_teacher, optimizer, samples, epoch, args, grad_sync=grad_sync, micro_step=it)

    def capture(example):
        return E.make_captured_step(model, model_ema, optimizer, args, example, epoch, grad_sync=grad_sync,
                                    fwd_bwd=step_forward_backward, extra={"model_teacher": model_teacher})

    return E.run_epoch(data_loader, optimizer, device, epoch, args, eager_step, capture, log_writer=log_writer,
                       print_freq=print_freq, loss_scale=loss_weights(epoch, args), model_key=model, use_graph=use_graph)
