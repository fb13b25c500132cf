"""Fine-tune / classification step: Point-MAE_SA3D/engine_finetune.py:70-176 (train_one_epoch),
:179-215 (evaluate) and the optimizer set-up of main_finetune.py:357-365 (layer-wise lr decay, util/lr_decay.py).

Per iteration, all on the GPU: FPS 8192 -> point_all (1200 for npoints 1024) with the HIP FPS kernel, ONE host-drawn
random subset of `npoints` of those indices shared by the batch (np.random.choice, like the reference), gather,
scale-and-translate augmentation, PointTransformer forward under bf16 autocast, cross-entropy, backward,
gradient clipping, AdamW.  The reference runs fp16 autocast + GradScaler; bf16 needs no loss scaling, so the
`loss_scaler` argument is accepted for signature compatibility and only its clip/step duty is performed.
"""
import math
import sys

import numpy as np
import torch

from . import ops
from . import streams
from .engine_pretrain import train_transforms

POINT_ALL = {1024: 1200, 2048: 2400, 4096: 4800, 8192: 8192}


# --------------------------------------------------------------------------- optimizer (util/lr_decay.py)
def get_layer_id_for_vit(name, num_layers):
    """P/util/lr_decay.py:65-78.  For this model: cls_token -> 0, blocks.blocks.{i}.* -> i + 1, everything else
    (encoder.*, pos_embed.*, cls_pos, norm_p.*, cls_head_finetune.*) -> num_layers."""
    if name in ("cls_token", "pos_embed") or name.startswith("patch_embed"):
        return 0
    if name.startswith("blocks"):
        return int(name.split(".")[2]) + 1
    return num_layers


def param_groups_lrd(model, weight_decay=0.05, no_weight_decay_list=(), layer_decay=0.75, num_layers=12):
    """Layer-wise lr decay groups (P/util/lr_decay.py:15-62): lr_scale = layer_decay ** (num_layers - layer_id);
    1-D parameters (and exact names in `no_weight_decay_list`) get weight_decay 0."""
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    groups = {}
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        no_decay = p.ndim == 1 or n in no_weight_decay_list
        lid = get_layer_id_for_vit(n, num_layers)
        key = "layer_%d_%s" % (lid, "no_decay" if no_decay else "decay")
        if key not in groups:
            groups[key] = {"lr_scale": scales[lid], "weight_decay": 0.0 if no_decay else weight_decay, "params": [],
                           "names": []}
        groups[key]["params"].append(p)
        groups[key]["names"].append(n)
    return list(groups.values())


def build_optimizer(model, lr, weight_decay=0.05, layer_decay=0.75, capturable=False, flat=False, max_norm=None,
                    num_layers=12):
    """torch.optim.AdamW(param_groups_lrd(...), lr) as in P/main_finetune.py:359-365 (multi-tensor fused update).
    capturable=True: per-group learning rates live in device tensors (hipGraph capture).
    flat=True: the same update (same groups: lr_scale per layer, no weight decay for 1-D parameters) as ONE pass over flat
    buffers -- optim.FlatAdamWEma with a per-element lr multiplier; gradient clipping (`max_norm`) happens inside its
    step, the learning rate is a device scalar (graph-capturable), and the GEMMs read the bf16 shadows it maintains."""
    if flat:
        from .optim import FlatAdamWEma
        scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
        return FlatAdamWEma(model, None, lr=lr, weight_decay=weight_decay, max_norm=max_norm if max_norm else 0.0,
                            no_decay_of=lambda n, p: p.ndim == 1,
                            lr_scale_of=lambda n: scales[get_layer_id_for_vit(n, num_layers)])
    groups = [{k: v for k, v in g.items() if k != "names"} for g in param_groups_lrd(
        model, weight_decay, no_weight_decay_list=[{"pos_embed", "cls_token"}], layer_decay=layer_decay)]
    dev = next(model.parameters()).device
    kw = {"fused": True} if dev.type == "cuda" else {}
    if capturable:
        kw["capturable"] = True
        for g in groups:
            g["lr"] = torch.tensor(float(lr), dtype=torch.float32, device=dev)
    return torch.optim.AdamW(groups, lr=lr, **kw)


def adjust_learning_rate(optimizer, epoch, args):
    """P/util/lr_sched.py:11-23 with the per-group lr_scale."""
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * \
            (1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))
    for g in optimizer.param_groups:
        v = lr * g["lr_scale"] if "lr_scale" in g else lr
        if torch.is_tensor(g["lr"]):
            g["lr"].fill_(v)                      # capturable optimizer: the captured update reads this tensor
        else:
            g["lr"] = v
    return lr


# --------------------------------------------------------------------------- the step
def sample_points(points, npoints, subset=None, rng=np.random):
    """P/engine_finetune.py:117-134: FPS to point_all, keep a random `npoints`-subset of the FPS order (the same
    subset for every cloud of the batch), gather.  `subset` injects the index draw (parity tests)."""
    if npoints not in POINT_ALL:
        raise NotImplementedError("npoints %d" % npoints)
    point_all = min(POINT_ALL[npoints], points.size(1))
    fps_idx = ops.furthest_point_sample(points.contiguous(), point_all)                 # (B, point_all) int32
    if subset is None:
        subset = rng.choice(point_all, npoints, False)
    if not torch.is_tensor(subset):                     # a device LongTensor is used as is (graph capture: static input)
        subset = torch.as_tensor(np.asarray(subset), dtype=torch.long).to(points.device)
    fps_idx = fps_idx.index_select(1, subset).contiguous()
    return ops.gather_operation(points.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()


def finetune_step(model, criterion, optimizer, points, targets, npoints=1024, max_norm=None, bf16=True, subset=None,
                  aug_draws=None, augment=True, update=True, accum_iter=1, presampled=False):
    """One iteration of P/engine_finetune.py:108-151.  points (B,N0,3) f32 and targets (B,) on the GPU.
    presampled=True: `points` already is the (B,npoints,3) output of sample_points (sampling overlapped elsewhere).
    -> {'loss', 'grad_norm', 'outputs'} as device tensors (no host sync)."""
    pts = points if presampled else sample_points(points, npoints, subset=subset)
    if augment:
        pts = train_transforms(pts, draws=aug_draws)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        outputs = model(pts)
        loss = criterion(outputs.float(), targets.long())
    (loss / accum_iter).backward()
    gnorm = None
    if update and hasattr(optimizer, "flat_grad_views"):      # FlatAdamWEma: clip + AdamW in one pass (its own max_norm)
        gnorm = optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    elif update:
        params = [p for p in model.parameters() if p.grad is not None]
        if max_norm is not None:
            gnorm = torch.nn.utils.clip_grad_norm_(params, max_norm, foreach=True)
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return {"loss": loss.detach(), "grad_norm": gnorm, "outputs": outputs.detach()}


class GraphedFinetuneStep:
    """The fine-tune iteration captured once as a hipGraph and replayed: at the reference's batch sizes (32-40 clouds) the
    step is a few hundred short launches and the host cannot keep the GPU fed.  Static inputs: the raw clouds, the
    labels and the random FPS subset (drawn on the host per call, copied in); the augmentation and DropPath/Dropout draws
    use the graph-safe device generator.  Needs an optimizer whose learning rates are device tensors that
    adjust_learning_rate fills in place: build_optimizer(..., capturable=True) or build_optimizer(..., flat=True).

    overlap_sampling=True: the point sampling (FPS 8192 -> point_all: one workgroup per cloud, 32-40 of the 256 CUs busy for
    >1 ms, a quarter of the step) is captured as its OWN graph and replayed on a second stream for the NEXT batch while the
    training graph of the current batch runs: call step(points, targets, next_points=<the following batch>).  Same
    arithmetic and the same order of host random draws as the single-graph form."""

    def __init__(self, model, criterion, optimizer, example_points, example_targets, npoints=1024, max_norm=None, bf16=True,
                 warmup_iters=3, rng=np.random, augment=True, overlap_sampling=False):
        """The `warmup_iters` un-captured iterations are real optimisation steps on the example batch (the allocator and
        the optimizer state must be warm before capture): pass warmup_iters=0 when the optimizer state already exists."""
        self.model, self.criterion, self.opt = model, criterion, optimizer
        self.npoints, self.max_norm, self.bf16, self.rng, self.augment = npoints, max_norm, bf16, rng, augment
        self.point_all = min(POINT_ALL[npoints], example_points.size(1))
        self.points = example_points.clone()
        self.targets = example_targets.clone()
        self.subset = torch.zeros(npoints, dtype=torch.long, device=example_points.device)
        self._pin = [torch.empty(npoints, dtype=torch.long).pin_memory() for _ in range(2)]
        self._pin_ev, self._pin_i = [None, None], 0
        self.overlap = bool(overlap_sampling)
        self._draw()
        if self.overlap:
            B = example_points.size(0)
            self.staged = torch.zeros(B, npoints, 3, dtype=torch.float32, device=example_points.device)
            self.pts = torch.zeros_like(self.staged)
            self.pts.copy_(sample_points(self.points, npoints, subset=self.subset))
            self.sstream = torch.cuda.Stream()
            self._ready, self._staged = torch.cuda.Event(), False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with streams.capture(self.graph):
            self.out = self._body()
        if self.overlap:
            self.sample_graph = torch.cuda.CUDAGraph()
            with streams.capture(self.sample_graph, stream=self.sstream):
                self.staged.copy_(sample_points(self.points, npoints, subset=self.subset))
            torch.cuda.synchronize()

    def _draw(self):
        """Host draw of the shared FPS subset -> device, through two alternating pinned buffers so that the copy is
        asynchronous (a pageable copy would make the host wait for everything queued before it, once per step)."""
        i, self._pin_i = self._pin_i, self._pin_i ^ 1
        if self._pin_ev[i] is not None:
            self._pin_ev[i].synchronize()
        self._pin[i].copy_(torch.from_numpy(self.rng.choice(self.point_all, self.npoints, False).astype(np.int64)))
        self.subset.copy_(self._pin[i], non_blocking=True)
        self._pin_ev[i] = torch.cuda.Event()
        self._pin_ev[i].record(torch.cuda.current_stream())

    def _body(self):
        if self.overlap:
            return finetune_step(self.model, self.criterion, self.opt, self.pts, self.targets, npoints=self.npoints,
                                 max_norm=self.max_norm, bf16=self.bf16, augment=self.augment, presampled=True)
        return finetune_step(self.model, self.criterion, self.opt, self.points, self.targets, npoints=self.npoints,
                             max_norm=self.max_norm, bf16=self.bf16, subset=self.subset, augment=self.augment)

    def _enqueue_sampling(self, points):
        """Sampler stream: (after everything the caller's stream has queued so far -- the producer of `points`, and the copy that
        consumed the previous staged batch) clouds in, subset draw, FPS + gather graph -> self.staged."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.sstream):
            self.sstream.wait_event(ev)
            self.points.copy_(points, non_blocking=True)
            points.record_stream(self.sstream)
            self._draw()
            self.sample_graph.replay()
            self._ready.record(self.sstream)
        self._staged = True

    def __call__(self, points, targets, next_points=None):
        if not self.overlap:
            self.points.copy_(points, non_blocking=True)
            self.targets.copy_(targets, non_blocking=True)
            self._draw()
            self.graph.replay()
            return self.out
        main = torch.cuda.current_stream()
        if not self._staged:                      # first call (or no look-ahead given last time): sample this batch now
            self._enqueue_sampling(points)
        main.wait_event(self._ready)
        self.pts.copy_(self.staged)
        self.targets.copy_(targets, non_blocking=True)
        self._staged = False
        if next_points is not None:               # queued BEFORE the training graph so that the two run side by side
            self._enqueue_sampling(next_points)
        self.graph.replay()
        return self.out


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler=None, max_norm=0,
                    mixup_fn=None, log_writer=None, args=None, npoints=0, print_freq=20, step=None):
    """Reference signature (P/engine_finetune.py:70-74).  The loss is read on the host every `print_freq` iterations
    (the reference syncs on loss.item() every iteration; a non-finite loss still stops the run at the next read).
    step: a GraphedFinetuneStep built on this model / optimizer -- the iteration is then a graph replay, and with
    overlap_sampling the loader is read one batch ahead so that the next batch's point sampling runs beside the current
    batch's training graph (accum_iter must be 1; a last batch of a different size runs eagerly)."""
    model.train(True)
    accum_iter = getattr(args, "accum_iter", 1)
    if step is not None and accum_iter != 1:
        raise NotImplementedError("graph replay with gradient accumulation")
    optimizer.zero_grad(set_to_none=True)
    n_iter = len(data_loader)
    seen, loss_sum, last = 0, 0.0, None

    def batches():          # (it, points, targets, next batch's points or None), everything already on the device
        prev = None
        for it, (_taxonomy_ids, _model_ids, data) in enumerate(data_loader):
            cur = (it, data[0].to(device, non_blocking=True), data[1].to(device, non_blocking=True))
            if prev is not None:
                yield prev + (cur[1],)
            prev = cur
        if prev is not None:
            yield prev + (None,)

    for it, points, targets, nxt in batches():
        if it % accum_iter == 0:
            adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        if step is not None and points.shape == step.points.shape:
            if step.overlap:
                out = step(points, targets, next_points=nxt if nxt is not None and nxt.shape == points.shape else None)
            else:
                out = step(points, targets)
        else:
            out = finetune_step(model, criterion, optimizer, points, targets, npoints=npoints,
                                max_norm=max_norm if max_norm else None, bf16=getattr(args, "bf16", True),
                                update=(it + 1) % accum_iter == 0, accum_iter=accum_iter)
        last = out["loss"]
        if (it + 1) % print_freq == 0 or it + 1 == n_iter:
            v = float(last)
            if not math.isfinite(v):
                print("Loss is {}, stopping training".format(v))
                sys.exit(1)
            loss_sum += v
            seen += 1
            if log_writer is not None:
                log_writer.add_scalar("loss", v, int((it / n_iter + epoch) * 1000))
    lrs = [float(g["lr"]) for g in optimizer.param_groups]
    return {"loss": loss_sum / max(seen, 1), "lr": max(lrs)}


@torch.no_grad()
def evaluate(data_loader, model, device, npoints=1024, bf16=True):
    """Top-1 accuracy / mean cross-entropy over a loader of (.., .., (points, label)) batches; test clouds are reduced
    to `npoints` by FPS alone (P/main_finetune.py validate: misc.fps(points, npoints))."""
    criterion = torch.nn.CrossEntropyLoss(reduction="sum")
    model.eval()
    n, correct, loss = 0, 0, 0.0
    for batch in data_loader:
        data = batch[-1]
        points, target = data[0].to(device), data[1].to(device).long().view(-1)
        if points.size(1) != npoints:
            points = ops.fps(points.contiguous(), npoints)[1]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            logits = model(points).float()
        loss += float(criterion(logits, target))
        correct += int((logits.argmax(-1) == target).sum())
        n += target.numel()
    return {"acc1": 100.0 * correct / max(n, 1), "loss": loss / max(n, 1), "n": n}
