"""Pretrain step engine: teacher -> mask -> student -> losses -> clip/AdamW -> EMA.

Host-side mirror of the reference's Point-MAE_SA3D/engine_pretrain.py::train_one_epoch (P/ below,
:38-271) with its collaborators: PointcloudScaleAndTranslate (P/datasets/data_transforms.py:20-35),
lr_sched.adjust_learning_rate (P/util/lr_sched.py:11-23), NativeScalerWithGradNormCount
(P/util/misc.py:250-276), the AdamW grouping rule (P/tools/builder.py:40-56), timm ModelEma and the
DDP wrap (P/main_pretrain_multi_gpu.py:296,309-311).

Same results, different execution: nothing in a step synchronises with the host (the reference
does ~650 tiny H2D/D2H transfers per step at B=128: SURVEY.md 3.1) -- augmentation and mask
generation are vectorised on the device, losses stay device tensors, the finite-loss guard is a
device flag read every `print_freq` steps, FPS/KNN run once and are shared by teacher and student
(identical samples), the teacher skips its unused reconstruction decoder, EMA and AdamW are
multi-tensor, and only the 36.8 M live gradients are all-reduced (bucketed, overlapped with backward).
"""
import math
import time
from contextlib import nullcontext

import torch
import torch.distributed as dist

from . import streams

# --------------------------------------------------------------------------- GEMM selection
def enable_tuned_gemms(tune_missing=False):
    """Point PyTorch's TunableOp at the hipBLASLt solution choices recorded for this step's GEMM shapes on gfx950
    (gm3d_amd/tuning/tunableop_gfx950_b128.csv, produced by running bench.py with PYTORCH_TUNABLEOP_TUNING=1).
    The step's GEMMs are small (M = 3200..8192 rows, K,N = 384..1536) or tall (262,144 rows): hipBLASLt's default
    heuristic leaves ~5 % of the step on the table for them.  Results recorded for another hipBLASLt/PyTorch
    version are ignored by TunableOp's validators; with tune_missing=True unknown shapes are tuned on first use."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950_b128.csv")
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(bool(tune_missing))
        ok = bool(tunable.read_file(path)) if os.path.exists(path) else False
        try:
            tunable.write_file_on_exit(False)     # never write into the working directory
        except Exception:
            pass
        return ok
    except Exception:   # TunableOp is an optimisation, never a requirement
        return False


# --------------------------------------------------------------------------- augmentation
class PointcloudScaleAndTranslate(object):
    """P/datasets/data_transforms.py:20-35, vectorised: one (B,3) scale ~ U[lo,hi] and one (B,3) shift ~
    U[-t,t] drawn on the device, applied in place.  `draws=(scale, shift)` injects the draws (parity tests)."""

    def __init__(self, scale_low=2.0 / 3.0, scale_high=3.0 / 2.0, translate_range=0.2):
        self.scale_low, self.scale_high, self.translate_range = scale_low, scale_high, translate_range

    def __call__(self, pc, draws=None, generator=None):
        B = pc.size(0)
        if draws is None and pc.is_cuda and pc.dtype == torch.float32 and pc.dim() == 3 and pc.size(2) == 3 and pc.is_contiguous():
            from ._capi import check, lib            # one launch after the draw (the same arithmetic, rounded the same way)
            from .ops import _ptr, _stream
            u = torch.rand(2, B, 3, device=pc.device, generator=generator)
            check(lib.gm3d_scale_translate(_ptr(pc), _ptr(u), self.scale_low, self.scale_high - self.scale_low, self.translate_range, B, pc.size(1),
                                           _stream()), "gm3d_scale_translate")
            return pc
        if draws is None:
            u = torch.rand(2, B, 3, device=pc.device, generator=generator)
            scale = u[0] * (self.scale_high - self.scale_low) + self.scale_low
            shift = (u[1] * 2 - 1) * self.translate_range
        else:
            scale, shift = (d.to(pc.device, pc.dtype) for d in draws)
        pc[:, :, 0:3] = pc[:, :, 0:3] * scale.unsqueeze(1) + shift.unsqueeze(1)
        return pc


train_transforms = PointcloudScaleAndTranslate()


# --------------------------------------------------------------------------- lr / optimizer / EMA
def adjust_learning_rate(optimizer, epoch, args):
    """Per-iteration linear warm-up then half-cycle cosine (P/util/lr_sched.py:11-23)."""
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * \
            (1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))
    for group in optimizer.param_groups:
        new = lr * group["lr_scale"] if "lr_scale" in group else lr
        if isinstance(group["lr"], torch.Tensor):
            group["lr"].fill_(new)  # capturable optimizers keep lr on the device
        else:
            group["lr"] = new
    return lr


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """P/tools/builder.py:40-56: no decay for 1-D params, *.bias and any name containing 'token'."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or "token" in name or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def ddp_segment(name):
    """Backward segment of a parameter of MaskedAutoencoderViT, in the order backward completes them: 0 = decoders, heads and
    mask token; 1 = the 12-block encoder stack (+ norm_p); 2 = token embed and positional embed."""
    if name.startswith("blocks.") or name.startswith("norm_p."):
        return 1
    if name.startswith("encoder.") or name.startswith("pos_embed."):
        return 2
    return 0


def build_optimizer(model, lr=1e-3, weight_decay=0.05, fused=True, capturable=False, flat=False, model_ema=None,
                    clip_grad=5.0, segment_of=None):
    """flat=True: FlatAdamWEma (gm3d_amd/optim.py) -- clip + AdamW + EMA + bf16 shadows in one pass over flat buffers;
    pass the ModelEma so the teacher's parameters join the layout.  Otherwise torch.optim.AdamW with the reference's
    parameter groups."""
    if flat:
        from .optim import FlatAdamWEma
        return FlatAdamWEma(model, model_ema, lr=lr, weight_decay=weight_decay, max_norm=clip_grad, segment_of=segment_of)
    groups = add_weight_decay(model, weight_decay=weight_decay)
    kw = {}
    if fused and next(model.parameters()).is_cuda:
        kw["fused"] = True
    if capturable:
        kw["capturable"] = True
        lr = torch.tensor(lr, dtype=torch.float32, device=next(model.parameters()).device)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, **kw)


class ModelEma:
    """timm-0.4.5 ModelEma: ema = deepcopy(model).eval(), update: v = v*decay + (1-decay)*model_v over every
    state-dict entry (floating entries as two multi-tensor launches; integer counters -- BatchNorm
    num_batches_tracked -- follow the same formula with truncation, like timm).  `decay` is a host float: a
    captured hipGraph bakes it in, so GraphedPretrainStep is re-captured when the epoch schedule changes it."""

    def __init__(self, model, decay=0.9999, device=""):
        import copy
        self.ema = copy.deepcopy(model)
        self.ema.eval()
        self.decay = decay
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._pairs = None
        self._int_flat = None

    def _build(self, model):
        msd = model.state_dict()
        needs_module = any(k.startswith("module.") for k in msd)
        fe, fm, ie, im = [], [], [], []
        skip = set(k for k, _ in self.ema.named_parameters()) if getattr(self, "params_in_optimizer", False) else ()
        for k, v in self.ema.state_dict().items():
            if k in skip:   # FlatAdamWEma updates the teacher's parameters inside its own kernel; buffers stay here
                continue
            mv = msd["module." + k if needs_module else k]
            (fe if v.dtype.is_floating_point else ie).append(v)
            (fm if v.dtype.is_floating_point else im).append(mv)
        self._int_flat = None
        if ie and ie[0].is_cuda:
            # the integer counters (BatchNorm num_batches_tracked, 0-dim int64) of each model become views of ONE tensor, so that
            # their update is 4 launches in all instead of 5 per counter; every in-place user (nn.BatchNorm1d, gm3d_bn_finalize,
            # load_state_dict, the buffer broadcast) keeps working on the views
            eb, mb = dict(self.ema.named_buffers()), {k[7:] if k.startswith("module.") else k: b for k, b in model.named_buffers()}
            keys = [k for k, b in eb.items() if b.dtype == torch.int64 and b.dim() == 0 and k in mb and mb[k].dtype == torch.int64]
            if len(keys) == len(ie):
                E, Mv = torch.stack([eb[k] for k in keys]), torch.stack([mb[k] for k in keys])
                for j, k in enumerate(keys):
                    eb[k].data, mb[k].data = E[j], Mv[j]
                self._int_flat, ie, im = (E, Mv), [], []
        self._pairs = (fe, fm, ie, im)

    def prepare(self, model):
        """settle the (teacher, student) buffer pairs before anything is captured (FlatAdamWEma's constructor calls this)"""
        if self._pairs is None:
            self._build(model)

    @torch.no_grad()
    def update(self, model):
        if self._pairs is None:
            # _build re-points the BatchNorm counters of BOTH models into one stacked tensor.  Inside a capture that is too late: graphs
            # captured earlier (the forward's counter increments) keep the old addresses, the old tensors are freed, and every replay
            # then adds 1 to whatever the allocator put there next (found in round 4 as a cloned loss growing by one ulp per replay).
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ModelEma.update() first called inside a hipGraph capture: call model_ema.prepare(model) (or run one "
                                   "eager step) before capturing")
            self._build(model)
        fe, fm, ie, im = self._pairs
        if fe:
            torch._foreach_lerp_(fe, fm, 1.0 - self.decay)  # v + (1-decay)*(m - v): one multi-tensor pass
        for e, m in zip(ie, im):
            e.copy_((e * self.decay + (1.0 - self.decay) * m).to(e.dtype))
        if self._int_flat is not None:
            E, Mv = self._int_flat
            if E.is_cuda:      # the same fp32 expression, truncated like the converting copy, in one launch instead of four
                from ._capi import lib, check
                from .ops import _ptr, _stream
                check(lib.gm3d_ema_counters(_ptr(E), _ptr(Mv), E.numel(), float(self.decay), 1.0 - float(self.decay), _stream()), "gm3d_ema_counters")
            else:
                E.copy_(E * self.decay + (1.0 - self.decay) * Mv)      # the same fp32 expression, truncated by the converting copy


def ema_decay_for_epoch(epoch):
    """P/engine_pretrain.py:55-60."""
    return 0.999 + epoch / 100 * (0.9999 - 0.999) if epoch < 100 else 0.9999


# --------------------------------------------------------------------------- data-parallel gradient sync
class GradSync:
    """Bucketed gradient all-reduce overlapped with backward (the DDP wrap of
    P/main_pretrain_multi_gpu.py:309-311, rebuilt for RCCL over xGMI).

    Gradients live as views into a few large flat buckets laid out in reverse parameter order
    (~ the order backward produces them).  A post-accumulate hook counts arrivals; when a bucket is
    complete its all-reduce is issued asynchronously on the process group's stream while backward keeps
    running.  xGMI is point-to-point, so a ring all-reduce is per-link bound: few large buckets (default
    32 MiB -> 5 collectives for the 147 MB of live fp32 gradients) beat DDP's 25 MB default + dead-parameter
    traffic.  Works on any backend (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, model=None, model_ema=None):
        """model / model_ema (optional): their parameters AND buffers are broadcast from rank 0 first, like the DDP constructor
        does (the reference seeds with args.seed + rank, P/main_pretrain_multi_gpu.py:175-176: without this every rank starts
        from different weights and averaging gradients never makes them equal again).  With only `params` given, those
        parameters are broadcast."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        if self.world > 1:
            if model is not None or model_ema is not None:
                broadcast_state(model, model_ema, None, group=process_group)
            else:
                _broadcast_tensors([p.data for p in self.params], group=process_group)
        self.buckets = []   # (flat, [params])
        self._owner = {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._seal(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._seal(cur)
        self._pending = [len(ps) for _, ps in self.buckets]
        self._works = []
        self._launched = [False] * len(self.buckets)
        # (one rank: SUM, see SegmentedDDPStep.__init__)
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl" and self.world > 1
        self.overlap = True   # False: hooks stay silent and finish() issues every bucket (used around captured graphs)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    @classmethod
    def from_flat(cls, optimizer, bucket_bytes=32 << 20, process_group=None):
        """Buckets = contiguous chunks of FlatAdamWEma's gradient buffer (its layout, not backward order): gradients are
        produced, all-reduced and consumed in place -- no flatten/unflatten copies.  Rank 0's parameters, optimizer state,
        EMA teacher and buffers are broadcast first (see __init__)."""
        self = cls.__new__(cls)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world > 1:
            broadcast_state(getattr(optimizer, "model", None), getattr(optimizer, "ema", None), optimizer, group=process_group)
        pairs = optimizer.flat_grad_views()
        self.params = [p for p, _ in pairs]
        self._views = pairs
        G = optimizer.G
        chunk = max(bucket_bytes // 4, 1)
        self.buckets, self._owner = [], {}
        bounds = list(range(0, G.numel(), chunk)) + [G.numel()]
        for b, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            self.buckets.append((G[lo:hi], []))
        for p, off in zip(optimizer._params, optimizer._offs):
            last = min((off + p.numel() - 1) // chunk, len(self.buckets) - 1)     # complete when its LAST element's chunk is
            self.buckets[last][1].append(p)
            self._owner[p] = last
        self._flat = G
        self._pending = [len(ps) for _, ps in self.buckets]
        self._works, self._launched = [], [False] * len(self.buckets)
        # (one rank: SUM, see SegmentedDDPStep.__init__)
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl" and self.world > 1
        self.overlap = False          # gradients reach the flat buffer by one gather at the end of backward
        return self

    def _seal(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._owner[p] = len(self.buckets)
        self.buckets.append((flat, ps))

    def zero_grad(self):
        if getattr(self, "_flat", None) is not None:
            self._flat.zero_()
            for p, g in self._views:
                p.grad = g
        else:
            for flat, _ in self.buckets:
                flat.zero_()
        self.reset()

    def reset(self):
        """Forget which buckets were issued: the next backward (or finish()) issues every bucket again."""
        self._pending = [len(ps) for _, ps in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _launch(self, b):
        self._launched[b] = True
        if self.world == 1:
            return
        flat = self.buckets[b][0]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append((dist.all_reduce(flat, op=op, group=self.group, async_op=True), b))

    def _hook(self, p):
        if not self.overlap:
            return
        b = self._owner[p]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b)

    def finish(self):
        """Call after backward: issues any bucket that never completed, waits, averages."""
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for work, b in self._works:
            work.wait()
            if not self._avg and self.world > 1:
                self.buckets[b][0].div_(self.world)
        # ready for the next step whether or not the caller goes through zero_grad() (the flat path zeroes through the
        # optimizer, and a replayed graph calls nothing at all): every finish() issues one collective per bucket
        self.reset()


def _broadcast_tensors(tensors, src=0, group=None):
    """Coalesced broadcast: one collective per dtype (pack, broadcast, unpack)."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def broadcast_state(model=None, model_ema=None, optimizer=None, src=0, group=None):
    """What the DDP constructor does at P/main_pretrain_multi_gpu.py:309-311, extended to everything a rank keeps: rank `src`'s
    student parameters and buffers (integer counters included), EMA teacher and -- for FlatAdamWEma -- the flat master / moment
    buffers and the step counter overwrite the other ranks'; the bf16 GEMM shadows are re-derived.  No-op without a group of >1."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ts = []
    flat = optimizer is not None and hasattr(optimizer, "flat_grad_views")
    if flat:
        ts += [optimizer.P, optimizer.M, optimizer.V, optimizer.step_dev]
        if optimizer.E is not None:
            ts.append(optimizer.E)
    seen = {t.data_ptr() for t in ts}
    lo, hi = (optimizer.P.data_ptr(), optimizer.P.data_ptr() + optimizer.P.numel() * 4) if flat else (0, 0)
    elo, ehi = (optimizer.E.data_ptr(), optimizer.E.data_ptr() + optimizer.E.numel() * 4) if flat and optimizer.E is not None else (0, 0)
    for mod in (model.module if hasattr(model, "module") else model, getattr(model_ema, "ema", model_ema)):
        if mod is None:
            continue
        for t in list(mod.parameters()) + list(mod.buffers()):
            a = t.data_ptr()
            if a in seen or lo <= a < hi or elo <= a < ehi:       # views of the flat buffers travel with them
                continue
            seen.add(a)
            ts.append(t.data)
    _broadcast_tensors(ts, src=src, group=group)
    if flat:
        optimizer.sync_shadows()


def broadcast_buffers(model, src=0, group=None):
    """DDP's default broadcast_buffers=True: rank 0's BatchNorm running statistics overwrite the other
    ranks' before the forward pass; one coalesced collective (3 launches: pack, broadcast, unpack)."""
    bufs = [b for b in model.buffers() if b.dtype.is_floating_point]
    if not bufs or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.broadcast(flat, src=src, group=group)
    views, off = [], 0
    for b in bufs:
        views.append(flat[off:off + b.numel()].view_as(b))
        off += b.numel()
    with torch.no_grad():
        torch._foreach_copy_(bufs, views)


def shard_for_rank(n_items, rank, world, epoch=0, seed=0, shuffle=True):
    """DistributedSampler-style partition (the reference omits the sampler: SURVEY.md 0.6): a seeded
    permutation per epoch, padded to a multiple of `world`, rank r takes items r, r+world, ..."""
    g = torch.Generator().manual_seed(seed + epoch)
    order = torch.randperm(n_items, generator=g) if shuffle else torch.arange(n_items)
    total = (n_items + world - 1) // world * world
    if total > n_items:
        order = torch.cat([order, order[: total - n_items]])
    return order[rank:total:world]


# --------------------------------------------------------------------------- the step
_arange_cache = {}


def _arange_ids(B, L, device):
    """(B,L) int64 rows of 0..L-1 (the all-visible id list of the teacher pass), cached per shape; created outside stream capture only
    (a tensor first made inside a capture lives in the graph's pool and holds nothing until the first replay)."""
    key = (B, L, device.type, device.index)
    t = _arange_cache.get(key)
    if t is None:
        t = torch.arange(L, device=device).unsqueeze(0).expand(B, L).contiguous()
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _arange_cache[key] = t
    return t


_const_cache = {}


def _const_scalar(device, value):
    """a cached f32 0-d constant on `device` (created outside stream capture only: see models_mae_learn_loss._zero_scalar)."""
    key = (str(device), float(value))
    c = _const_cache.get(key)
    if c is None:
        c = torch.full((), float(value), dtype=torch.float32, device=device)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _const_cache[key] = c
    return c


_false_cache = {}


def _all_false(B, L, device):
    """(B,L) bool zeros (the teacher's all-visible mask), cached per shape like _arange_ids; created outside stream capture only."""
    key = (B, L, device.type, device.index)
    t = _false_cache.get(key)
    if t is None:
        t = torch.zeros(B, L, dtype=torch.bool, device=device)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _false_cache[key] = t
    return t


def backward_and_collect(total, raw, optimizer, grad_sync, accum=1, accum_first=True, accum_last=True, async_w=True):
    """zero (first micro-batch of the window) -> backward -> gradients where the update / all-reduce expects them.
    total: the scalar objective (already divided by accum), or a tuple of scalar losses whose sum / accum is the objective.
    Shared by this engine and engine_pretrain_Classifier_SVM."""
    flat_opt = optimizer is not None and hasattr(optimizer, "flat_grad_views")
    flat_sync = grad_sync is not None and getattr(grad_sync, "_flat", None) is not None and flat_opt
    if grad_sync is not None and not flat_sync:
        if accum_first:
            grad_sync.zero_grad()
        grad_sync.overlap = grad_sync.overlap and accum_last     # bucket collectives only on the window's summed gradients
    elif flat_opt:
        optimizer.zero_grad(set_to_none=True)      # every micro-batch: backward writes fresh tensors / its flat slots
    elif accum_first:
        if optimizer is not None:
            optimizer.zero_grad(set_to_none=True)
        else:
            for p in raw.parameters():
                p.grad = None
    from .fused import async_wgrad
    # the encoder stack's weight-gradient GEMMs beside the embed's backward; joined on exit
    roots = list(total) if isinstance(total, (tuple, list)) else None
    dev = roots[0].device if roots else total.device
    with (async_wgrad(dev) if async_w else nullcontext()):
        if roots:     # several scalar losses whose sum / accum is the objective: seeded with a cached constant 1 / accum each
            seed = _const_scalar(dev, 1.0 / accum)
            torch.autograd.backward(roots, [seed] * len(roots))
        else:
            total.backward()
    if flat_opt and (flat_sync or accum > 1):
        optimizer.gather_grads()   # one multi-tensor copy into the flat buffer the all-reduce works on
        if accum > 1:
            optimizer.accumulate(last=accum_last)


def step_forward_backward(model, model_ema, samples, epoch, args, grad_sync=None, mask_noise=None, augment=True,
                          aug_draws=None, optimizer=None, accum_first=True, accum_last=True):
    """First half of P/engine_pretrain.py:77-197: augment -> teacher -> mask -> student -> losses -> backward.
    Leaves the gradients in p.grad (views of GradSync's flat buckets when data-parallel).
    accum_first / accum_last: position of this micro-batch in its accumulation window (args.accum_iter > 1): gradients are
    zeroed only before the first micro-batch and summed over the window (P/engine_pretrain.py:195-212 with update_grad =
    (it + 1) % accum_iter == 0); the data-parallel collectives run once, on the sums, in the last one."""
    raw = model.module if hasattr(model, "module") else model
    teacher = model_ema.ema
    L = raw.num_group
    len_keep = int(L * (1 - args.mask_ratio))
    if augment:
        samples = train_transforms(samples, draws=aug_draws)
    if getattr(args, "bf16", False):
        from .fused import weight_cache
        weight_cache.pin(raw)        # no-ops after the first call
        weight_cache.pin(teacher)
        weight_cache.refresh()       # ONE multi-tensor cast of all GEMM weights per model per step
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if getattr(args, "bf16", False) else nullcontext()
    B = samples.shape[0]
    visible_mask = _all_false(B, L, samples.device)      # read only
    with amp:
        with torch.no_grad():
            group = teacher.group_divider(samples)  # FPS + KNN once; shared with the student
        with torch.no_grad():
            all_ids = (_arange_ids(B, L, samples.device), _arange_ids(B, 0, samples.device))
            outs_ema = teacher(samples, mask=visible_mask, num_visible=L, group=group, need_pix_pred=False, ids=all_ids)
            ids = None
            if samples.is_cuda and L <= 64:       # mask (f32 and bool) + visible / masked id lists in one launch
                mask, vis_ids, mask_ids, bool_masked_pos = teacher.generate_mask_ids(
                    outs_ema["loss_pred"], mask_ratio=args.mask_ratio, guide=True, epoch=epoch, total_epoch=args.epochs, noise=mask_noise,
                    want_bool=True)
                ids = (vis_ids, mask_ids)
            else:
                mask = teacher.generate_mask(outs_ema["loss_pred"], mask_ratio=args.mask_ratio, guide=True, epoch=epoch,
                                             total_epoch=args.epochs, noise=mask_noise)
                bool_masked_pos = mask.flatten(1).to(torch.bool)
        outs = model(samples, mask=bool_masked_pos, num_visible=len_keep, group=group, ids=ids)
        M = outs["mask_num"]
        # the two losses read the last M tokens of the full predictions: passed whole (full_pred=) so that the fused losses fold the
        # slice and its zero-filling backward in
        loss_outs = raw.forward_loss(outs["pix_pred"][:, -M:], outs["neighborhood"], outs["mask"],
                                     mask_ids=ids[1] if ids is not None else None, full_pred=outs["pix_pred"])
        loss_mse, loss_chfr = loss_outs["MSE_mean"], loss_outs["Chamfer_mean"]
        # P/:153: 13.889 * MSE + 1.0 * Chamfer; MSE is identically 0 in this variant (no launches, forward or backward, for it)
        loss = loss_chfr if loss_outs.get("MSE_zero") else 13.889 * loss_mse + 1.0 * loss_chfr
        loss_learn = raw.forward_learning_loss(outs["loss_pred"][:, -M:], bool_masked_pos,
                                               loss_outs["matrix"].detach(), relative=args.relative, full_pred=outs["loss_pred"])
    accum = getattr(args, "accum_iter", 1)
    # P/:190,195: (loss + loss_learn) / accum_iter -> backward.  Two roots seeded with the constant 1 / accum_iter: the same gradients
    # without the sum, the division and the ones_like launch
    backward_and_collect((loss, loss_learn), raw, optimizer, grad_sync, accum, accum_first, accum_last)
    return {"loss": loss.detach(), "loss_learn": loss_learn.detach(), "loss_chfr": loss_chfr.detach(),
            "loss_mse": loss_mse.detach(), "mask": bool_masked_pos, "matrix": loss_outs["matrix"].detach(),
            "teacher_loss_pred": outs_ema["loss_pred"]}


def step_update(model, model_ema, optimizer, clip_grad=5.0):
    """Second half (P/util/misc.py:262-266 + P/engine_pretrain.py:208-212): clip -> AdamW -> EMA."""
    raw = model.module if hasattr(model, "module") else model
    if hasattr(optimizer, "flat_grad_views"):      # FlatAdamWEma: clip + AdamW + EMA(params) + bf16 shadows, one pass
        grad_norm = optimizer.step()
        model_ema.update(raw)                      # BatchNorm buffers only
        return grad_norm
    grad_norm = torch.nn.utils.clip_grad_norm_(raw.parameters(), clip_grad, foreach=True)
    optimizer.step()
    model_ema.update(raw)
    return grad_norm


def pretrain_step(model, model_ema, optimizer, samples, epoch, args, grad_sync=None, mask_noise=None,
                  augment=True, aug_draws=None, clip_grad=5.0, micro_step=None):
    """One iteration of P/engine_pretrain.py:77-212.  `samples` (B,N,3) f32 on the GPU (modified in
    place by the augmentation, like the reference).  Returns device tensors only -- no host sync.
    micro_step: the iteration index `data_iter_step` when args.accum_iter > 1 -- gradients are summed over accum_iter
    consecutive iterations and clip / AdamW / EMA / zero_grad run only when (micro_step + 1) % accum_iter == 0
    (P/engine_pretrain.py:196-212, P/util/misc.py:256-270); the other iterations return grad_norm = None like the reference."""
    accum = getattr(args, "accum_iter", 1)
    if accum > 1 and micro_step is None:
        raise ValueError("args.accum_iter = %d needs micro_step (the iteration index) to place the update" % accum)
    first = accum == 1 or micro_step % accum == 0
    last = accum == 1 or (micro_step + 1) % accum == 0
    if grad_sync is not None:
        overlap_was = grad_sync.overlap
        raw = model.module if hasattr(model, "module") else model
        broadcast_buffers(raw, group=grad_sync.group)          # DDP's per-forward buffer broadcast from rank 0
    out = step_forward_backward(model, model_ema, samples, epoch, args, grad_sync=grad_sync, mask_noise=mask_noise,
                                augment=augment, aug_draws=aug_draws, optimizer=optimizer, accum_first=first, accum_last=last)
    if grad_sync is not None:
        grad_sync.overlap = overlap_was
    if not last:
        out["grad_norm"] = None
        return out
    if grad_sync is not None:
        grad_sync.finish()
    out["grad_norm"] = step_update(model, model_ema, optimizer, clip_grad)
    return out


class GraphedPretrainStep:
    """The whole pretrain step captured once as a hipGraph and replayed: the step is ~700 launches, so the host cannot keep the
    GPU fed in eager mode.  Everything that changes between steps is device state the captured kernels read: the input batch
    (copied into a static buffer), the learning rate (a device tensor), RNG offsets (graph-safe philox).  `epoch` (the mask
    schedule's len_loss) and the EMA decay are baked in at capture: re-capture when the epoch changes (train_one_epoch does).

    Layouts: ONE graph (single GPU, accum_iter 1); TWO graphs (forward+backward | clip+AdamW+EMA) when something eager has to
    run between them -- the bucketed all-reduce of the flat gradient buffer (`grad_sync`) and/or gradient accumulation
    (args.accum_iter > 1: every call replays the first graph, which ends with GA += G; calls with update=True then move the sum
    back, all-reduce it and replay the second).

    CAUTION (ROCm 7.2 / torch 2.10): PyTorch's multi-block reduce_kernel (sum/mean over >~64k elements, column
    sums over thousands of rows) returns stale results from the second replay on when graph-pool memory is
    reused -- reproduced without any of our code by tools/graph_reduce_test2.py.  The step is only replay-safe
    where every such reduction is one of our own kernels; tests/test_gpu_graph.py compares replay against eager."""

    def __init__(self, model, model_ema, optimizer, args, example, epoch, warmup_iters=3, augment=True,
                 inject_mask_noise=False, grad_sync=None, fwd_bwd=None, extra=None):
        """fwd_bwd / extra: another engine's forward+backward half with its extra keyword arguments (the published-run
        variant passes its own step_forward_backward and the frozen teacher); default: this module's.
        inject_mask_noise=True: the (B,L) ranking noise of generate_mask becomes a static input filled by the caller
        (deterministic replays for the tests); otherwise it is drawn inside the graph.
        warmup_iters: eager iterations on `example` before the capture -- they ARE optimizer steps; pass 0 when the caller has
        already run eager iterations (train_one_epoch does: the first iterations of the first epoch are the warm-up)."""
        self.static_in = example.clone()
        L = (model.module if hasattr(model, "module") else model).num_group
        self.static_noise = torch.rand(example.shape[0], L, device=example.device) if inject_mask_noise else None
        self.model, self.ema, self.opt, self.args, self.epoch = model, model_ema, optimizer, args, epoch
        self.grad_sync = grad_sync
        self.accum = getattr(args, "accum_iter", 1)
        if self.accum > 1 and not hasattr(optimizer, "accumulate"):
            raise NotImplementedError("captured gradient accumulation needs the flat optimizer (build_optimizer(flat=True))")
        kw = dict(augment=augment, mask_noise=self.static_noise, grad_sync=grad_sync, **(extra or {}))
        if self.accum > 1:
            kw.update(accum_first=True, accum_last=False)      # uniform micro-batch: GA += G (see __call__)
            if optimizer.GA is None:                           # persistent memory, never the graph's pool
                optimizer.GA = torch.zeros_like(optimizer.G)
        fwd_bwd = fwd_bwd or step_forward_backward
        if grad_sync is not None:
            grad_sync.overlap = False
        two = grad_sync is not None or self.accum > 1

        def whole(samples):
            out = fwd_bwd(model, model_ema, samples, epoch, args, optimizer=optimizer, **kw)
            if self.accum > 1:
                self._collect()
            if grad_sync is not None:
                grad_sync.finish()
            out["grad_norm"] = step_update(model, model_ema, optimizer)
            return out

        if warmup_iters:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup_iters):
                    whole(self.static_in.clone())
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread polls events; only the capturing thread's calls may abort a capture
        import os
        mode = "thread_local" if dist.is_initialized() else "global"
        if not two:
            self.graph2 = None
            with streams.capture(self.graph, capture_error_mode=mode):
                self.out = whole(self.static_in)
        else:
            with streams.capture(self.graph, capture_error_mode=mode):
                self.out = fwd_bwd(model, model_ema, self.static_in, epoch, args, optimizer=optimizer, **kw)
            self.graph2 = torch.cuda.CUDAGraph()
            if hasattr(optimizer, "mark_grads_filled"):
                optimizer.mark_grads_filled()       # G is complete when this graph runs (gathered / collected / all-reduced)
            with streams.capture(self.graph2, pool=self.graph.pool(), capture_error_mode=mode):
                self.out["grad_norm"] = step_update(model, model_ema, optimizer)

    def _collect(self):
        """accumulated sum -> G (the buffer the all-reduce and the update work on); GA cleared for the next window."""
        self.opt.G.copy_(self.opt.GA)
        self.opt.GA.zero_()
        self.opt.mark_grads_filled()

    def __call__(self, samples, mask_noise=None, update=True):
        """update=False (gradient accumulation, not the last micro-batch of its window): forward + backward only; the returned
        grad_norm is None, like the reference's loss_scaler(update_grad=False)."""
        self.static_in.copy_(samples, non_blocking=True)
        if self.static_noise is not None:
            self.static_noise.copy_(mask_noise, non_blocking=True)
        if self.grad_sync is not None and getattr(self, "model", None) is not None:
            broadcast_buffers(self.model.module if hasattr(self.model, "module") else self.model,
                              group=self.grad_sync.group)   # like DDP, every forward
        self.graph.replay()
        if self.graph2 is None:
            return self.out
        if not update:
            out = dict(self.out)
            out["grad_norm"] = None
            return out
        if getattr(self, "accum", 1) > 1:
            self._collect()
        if self.grad_sync is not None:
            self.grad_sync.finish()
        self.graph2.replay()
        return self.out


class SegmentedDDPStep:
    """Data-parallel pretrain step with the gradient all-reduce hidden behind backward, as FOUR hipGraphs with eager RCCL
    collectives between them (no collective is captured):

        graph 1  forward + backward of losses, heads and both decoders   -> all-reduce (async) of segment 0's gradients
        graph 2  backward of the 12-block encoder stack                  -> all-reduce (async) of segment 1's gradients
        graph 3  backward of token embed + positional embed              -> all-reduce of segment 2 (+ every non-decayed
                                                                            parameter: biases, norms, tokens)
        graph 4  clip + AdamW + EMA (after the collectives)

    The backward is cut with torch.autograd.grad at the tensors where the segments meet (x_vis / pos_full: encoder output and
    decoder positions; tokens / pos_all: embed outputs), so the sum of the three partial backwards is exactly the single
    backward -- tests/test_gpu_graph.py checks it against the un-segmented step.  Needs FlatAdamWEma built with
    segment_of=ddp_segment: each segment's weight gradients are one contiguous range of the flat buffer.
    While graph 2 runs, RCCL's stream reduces segment 0 (57 MB); while graph 3 runs, segment 1 (85 MB); only segment 2's
    ~5 MB collective is exposed.
    Construction broadcasts rank 0's parameters, optimizer state, EMA teacher and buffers (DDP's constructor does the same for
    the model; the reference seeds every rank differently, P/main_pretrain_multi_gpu.py:175-176).
    args.accum_iter > 1: the segments' gradients are summed over the window in a second flat buffer and the three collectives
    run once, on the sums, after the window's last backward (no overlap in that mode)."""

    def __init__(self, model, model_ema, optimizer, args, example, epoch, warmup_iters=3, augment=True, inject_mask_noise=False,
                 process_group=None, use_graphs=True, broadcast=True):
        assert getattr(optimizer, "segment_ranges", None) is not None, "build the optimizer with segment_of=ddp_segment"
        self.model, self.ema, self.opt, self.args, self.epoch = model, model_ema, optimizer, args, epoch
        self.raw = model.module if hasattr(model, "module") else model
        self.group, self.augment = process_group, augment
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # RCCL averages inside its ring kernel (ReduceOp.AVG).  A group of ONE rank keeps SUM: AVG there runs RCCL's one-rank
        # pre-multiply kernel over the whole range (71 us per collective measured, 0.21 ms per step: tools/seg_timeline.py), which
        # no N > 1 run executes -- a one-rank rehearsal of the layout should not be charged for it.
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl" and self.world > 1
        self.accum = getattr(args, "accum_iter", 1)
        if broadcast:
            broadcast_state(self.raw, model_ema, optimizer, group=process_group)
        L = self.raw.num_group
        self.static_in = example.clone()
        self.static_noise = torch.rand(example.shape[0], L, device=example.device) if inject_mask_noise else None
        views = dict((id(p), g) for p, g in optimizer.flat_grad_views())
        self.seg_params = {sg: [p for p in ps] for sg, ps in optimizer.segment_params.items()}
        self.seg_views = {sg: [views[id(p)] for p in ps] for sg, ps in self.seg_params.items()}
        self.use_graphs = use_graphs
        self.graphs = None
        if use_graphs:
            if warmup_iters:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(warmup_iters):
                        self._eager(self.static_in.clone())
                torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            import os
            mode = "thread_local" if dist.is_initialized() else "global"
            self.graphs = [torch.cuda.CUDAGraph() for _ in range(4)]
            with streams.capture(self.graphs[0], capture_error_mode=mode):
                self.out = self._phase1(self.static_in)
            for k, phase in ((1, self._phase2), (2, self._phase3)):
                with streams.capture(self.graphs[k], pool=self.graphs[0].pool(), capture_error_mode=mode):
                    phase()
            with streams.capture(self.graphs[3], pool=self.graphs[0].pool(), capture_error_mode=mode):
                self.opt.mark_grads_filled()
                self.out["grad_norm"] = step_update(model, model_ema, optimizer)

    # ---- the three backward segments --------------------------------------------------------------------------------
    def _store(self, seg, grads):
        """gradients of one segment -> their slots of the flat buffer (ONE multi-tensor copy; unused parameters: zeros)."""
        views, have, zero = self.seg_views[seg], [], []
        for v, g in zip(views, grads):
            if g is not None and g.data_ptr() == v.data_ptr():
                continue                      # written into its slot by the backward node itself (fused._wgrad_batched)
            (have if g is not None else zero).append((v, g))
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if zero:
            torch._foreach_zero_([v for v, _ in zero])

    def _phase1(self, samples):
        raw, teacher, args, epoch = self.raw, self.ema.ema, self.args, self.epoch
        self.opt.zero_grad(set_to_none=True)       # no stale p.grad may be gathered over the segment gradients by step()
        L = raw.num_group
        len_keep = int(L * (1 - args.mask_ratio))
        if self.augment:
            samples = train_transforms(samples)
        bf16 = getattr(args, "bf16", False)
        if bf16:
            from .fused import weight_cache
            weight_cache.pin(raw)
            weight_cache.pin(teacher)
            weight_cache.refresh()
        amp = torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else nullcontext()
        B = samples.shape[0]
        visible_mask = _all_false(B, L, samples.device)      # read only
        with amp:
            with torch.no_grad():
                group = teacher.group_divider(samples)
                all_ids = (_arange_ids(B, L, samples.device), _arange_ids(B, 0, samples.device))
                outs_ema = teacher(samples, mask=visible_mask, num_visible=L, group=group, need_pix_pred=False, ids=all_ids)
                mask, vis_ids, mask_ids, bool_masked_pos = teacher.generate_mask_ids(
                    outs_ema["loss_pred"], mask_ratio=args.mask_ratio, guide=True, epoch=epoch, total_epoch=args.epochs, noise=self.static_noise,
                    want_bool=True)
            from . import models_mae_learn_loss as MM
            vis_only = MM.VISIBLE_EMBED and raw.encoder.fused(group[0]) and vis_ids.shape[1] < L
            # segment 2 | segment 1 boundary: the encoder sees detached leaves (the visible tokens only, when the embed can stop there)
            tokens = raw.encoder(group[0], vis_ids=vis_ids if vis_only else None)
            pos_all = raw.embed_pos(group[1])
            tokens_d, pos_all_d = tokens.detach().requires_grad_(True), pos_all.detach().requires_grad_(True)
            outs = self.model(samples, mask=bool_masked_pos, num_visible=len_keep, group=group, tokens=tokens_d, pos_all=pos_all_d,
                              ids=(vis_ids, mask_ids), cut=True, tokens_visible=vis_only)
            M = outs["mask_num"]
            loss_outs = raw.forward_loss(outs["pix_pred"][:, -M:], outs["neighborhood"], outs["mask"], mask_ids=mask_ids,
                                         full_pred=outs["pix_pred"])
            loss_mse, loss_chfr = loss_outs["MSE_mean"], loss_outs["Chamfer_mean"]
            loss = loss_chfr if loss_outs.get("MSE_zero") else 13.889 * loss_mse + 1.0 * loss_chfr
            loss_learn = raw.forward_learning_loss(outs["loss_pred"][:, -M:], bool_masked_pos, loss_outs["matrix"].detach(),
                                                   relative=args.relative, full_pred=outs["loss_pred"])
        accum = getattr(args, "accum_iter", 1)
        x_vis_d, pos_full_d = outs["features"], outs["pos_full"]   # segment 1 | segment 0 boundary (detached leaves)
        p0 = self.seg_params[0]
        seed = _const_scalar(loss.device, 1.0 / accum)             # two roots seeded with 1 / accum: no sum, division, ones_like
        g = torch.autograd.grad([loss, loss_learn], p0 + [x_vis_d, pos_full_d], grad_outputs=[seed, seed], allow_unused=True)
        self._store(0, g[:len(p0)])
        self._cut1 = outs["cut"] + (g[len(p0)], g[len(p0) + 1])
        self._cut2 = (tokens, pos_all, tokens_d, pos_all_d)
        return {"loss": loss.detach(), "loss_learn": loss_learn.detach(), "loss_chfr": loss_chfr.detach(),
                "loss_mse": loss_mse.detach(), "mask": bool_masked_pos, "matrix": loss_outs["matrix"].detach(),
                "teacher_loss_pred": outs_ema["loss_pred"]}

    def _phase2(self):
        x_vis, pos_full, gx, gp = self._cut1
        tokens_d, pos_all_d = self._cut2[2], self._cut2[3]
        p1 = self.seg_params[1]
        outs = [t for t, gt in ((x_vis, gx), (pos_full, gp)) if gt is not None]
        gouts = [gt for gt in (gx, gp) if gt is not None]
        g = torch.autograd.grad(outs, p1 + [tokens_d, pos_all_d], grad_outputs=gouts, allow_unused=True)   # (no fused.async_wgrad
        # region around a segment: measured 7.98 vs 7.87 ms with one around this one, profiles/NEGATIVE_RESULTS.md round 4)
        self._store(1, g[:len(p1)])
        self._cut3 = (g[len(p1)], g[len(p1) + 1])

    def _phase3(self):
        tokens, pos_all = self._cut2[0], self._cut2[1]
        gt, gp = self._cut3
        p2 = self.seg_params[2]
        outs = [t for t, q in ((tokens, gt), (pos_all, gp)) if q is not None]
        gouts = [q for q in (gt, gp) if q is not None]
        g = torch.autograd.grad(outs, p2, grad_outputs=gouts, allow_unused=True)
        self._store(2, g)

    def _reduce(self, seg):
        """async all-reduce (mean) of one segment's range of the flat gradient buffer on the process group's stream."""
        if not dist.is_initialized():
            return None
        lo, hi = self.opt.segment_ranges[seg]
        buf = self.opt.G[lo:hi]
        w = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
        return (w, buf)

    def _wait(self, works):
        for item in works:
            if item is None:
                continue
            w, buf = item
            w.wait()
            if not self._avg and self.world > 1:
                buf.div_(self.world)

    def _run(self, phases, update):
        """phases: three callables (graph replays or the eager segment functions).  accum_iter == 1: each segment's collective is
        issued right behind its phase; accum_iter > 1: gradients join the window's sum, collectives only when `update`."""
        if self.accum == 1:
            works = []
            for k in range(3):
                phases[k]()
                works.append(self._reduce(k))
            self._wait(works)
            return True
        for k in range(3):
            phases[k]()
        self.opt.accumulate(last=update)
        if update:
            self._wait([self._reduce(k) for k in range(3)])
        return update

    def _eager(self, samples, update=True):
        res = {}
        done = self._run([lambda: res.update(self._phase1(samples)), self._phase2, self._phase3], update)
        self._cut1 = self._cut2 = self._cut3 = None      # eager: let the autograd graph go (captured graphs keep theirs alive)
        if done:
            self.opt.mark_grads_filled()
            res["grad_norm"] = step_update(self.model, self.ema, self.opt)
        else:
            res["grad_norm"] = None
        return res

    def __call__(self, samples, mask_noise=None, update=True):
        if self.static_noise is not None:
            self.static_noise.copy_(mask_noise, non_blocking=True)
        broadcast_buffers(self.raw, group=self.group)      # DDP's per-forward BatchNorm-buffer broadcast from rank 0
        if self.graphs is None:
            return self._eager(samples, update)
        self.static_in.copy_(samples, non_blocking=True)
        if self._run([g.replay for g in self.graphs[:3]], update):
            self.graphs[3].replay()
            return self.out
        out = dict(self.out)
        out["grad_norm"] = None
        return out


# --------------------------------------------------------------------------- the epoch loop
import weakref
_warm = weakref.WeakKeyDictionary()   # model -> eager iterations run so far (lazy initialisation, library workspaces, code
#                                       objects); weak keys: a new model at a recycled address starts from zero again
EAGER_WARMUP_ITERS = 3


def make_captured_step(model, model_ema, optimizer, args, example, epoch, grad_sync=None, fwd_bwd=None, extra=None):
    """The captured form of one iteration of this epoch for the process's situation: SegmentedDDPStep when a process group of
    more than one rank is alive and the flat optimizer is laid out by backward segment, else GraphedPretrainStep (two graphs
    around the bucketed all-reduce when data-parallel, one graph on a single GPU).  No warm-up iterations: the caller has run
    eager ones."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1 and fwd_bwd is None and getattr(optimizer, "segment_ranges", None) is not None:
        return SegmentedDDPStep(model, model_ema, optimizer, args, example, epoch, warmup_iters=0,
                                process_group=getattr(grad_sync, "group", None), broadcast=False)
    return GraphedPretrainStep(model, model_ema, optimizer, args, example, epoch, warmup_iters=0, grad_sync=grad_sync,
                               fwd_bwd=fwd_bwd, extra=extra)


def run_epoch(data_loader, optimizer, device, epoch, args, eager_step, capture, log_writer=None, print_freq=20,
              loss_scale=(13.889, 1.0), model_key=None, use_graph=True):
    """The loop of P/engine_pretrain.py:68-257 around one engine's iteration (shared with engine_pretrain_Classifier_SVM).
    eager_step(samples, it) -> out dict; capture(example) -> a callable step(samples, update=) or None.
    Iterations run as hipGraph replays (what bench.py measures) once EAGER_WARMUP_ITERS eager iterations have run in this
    process -- the first iterations of the first epoch; a batch of another shape (a smaller last batch) runs eagerly.
    Learning rate: per iteration, set only at the start of an accumulation window (P/:72-73)."""
    n_iter = len(data_loader)
    accum = getattr(args, "accum_iter", 1)
    if model_key is None:          # a weak dictionary cannot key on None: the warm-up count then belongs to the optimizer
        model_key = optimizer
    # P/engine_pretrain.py:62 `optimizer.zero_grad()` at the start of every epoch: micro-batches of a window the previous epoch
    # left unfinished (len(data_loader) % accum_iter != 0) are discarded, not added to this epoch's first update
    if getattr(optimizer, "GA", None) is not None:
        optimizer.GA.zero_()
    if optimizer is not None and accum > 1:
        optimizer.zero_grad()
    sums, n_upd, gsum, lr = None, 0, None, 0.0
    bad = torch.zeros((), dtype=torch.bool, device=device)
    t0, seen = time.time(), 0
    graphed, t_replay, seen_replay, n_replay, t_capture = None, 0.0, 0, 0, 0.0
    rank0 = not dist.is_initialized() or dist.get_rank() == 0
    for it, points in enumerate(data_loader):
        if it % accum == 0:
            lr = adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        samples = points.to(device, non_blocking=True)
        last = (it + 1) % accum == 0
        if (use_graph and graphed is None and it % accum == 0 and _warm.get(model_key, 0) >= EAGER_WARMUP_ITERS * accum
                and samples.is_cuda):
            tc = time.time()
            graphed = capture(samples)
            torch.cuda.synchronize()
            t_capture = time.time() - tc
            use_graph = graphed is not None
            tr0 = time.time()
        if graphed is not None and samples.shape == graphed.static_in.shape:
            out = graphed(samples, update=last)
            seen_replay += samples.shape[0]
            n_replay += 1
        else:
            out = eager_step(samples, it)
            _warm[model_key] = _warm.get(model_key, 0) + 1
        vec = torch.stack([out["loss"] + out["loss_learn"], out["loss_learn"], out["loss_mse"] * loss_scale[0],
                           out["loss_chfr"] * loss_scale[1]])
        bad |= ~torch.isfinite(vec).all()
        sums = vec if sums is None else sums + vec
        if out["grad_norm"] is not None:        # the reference's grad_norm is None on accumulation-only iterations
            g = out["grad_norm"].float().reshape(())
            bad |= ~torch.isfinite(g)
            gsum = g.clone() if gsum is None else gsum + g
            n_upd += 1
        seen += samples.shape[0]
        if (it + 1) % print_freq == 0 or it + 1 == n_iter:
            if bool(bad):  # the reference exits on a non-finite loss (:173-175,185-187); one sync per print_freq
                raise FloatingPointError("non-finite loss in epoch %d near iteration %d" % (epoch, it))
            cur = (sums / (it + 1)).tolist() + [float(gsum) / max(n_upd, 1) if gsum is not None else 0.0]
            if log_writer is not None:
                step = n_iter * epoch + it
                for name, v in zip(("train_loss", "train_loss_learn", "train_loss_MSE", "train_loss_Chfr", "grad_norm"), cur):
                    log_writer.add_scalar(name, v, step)
                log_writer.add_scalar("lr", lr, step)
            if rank0:
                print("Epoch: [%d]  [%d/%d]  lr %.6f  loss %.4f  loss_learn %.4f  loss_mse %.4f  loss_chfr %.4f  grad_norm %.3f  "
                      "%.0f clouds/s" % (epoch, it + 1, n_iter, lr, cur[0], cur[1], cur[2], cur[3], cur[4],
                                         seen / (time.time() - t0)))
        if it + 1 >= n_iter:
            break
    if device.type == "cuda":
        torch.cuda.synchronize()
    t_end = time.time()
    if graphed is not None:
        t_replay = t_end - tr0
    stats = torch.cat([sums / max(n_iter, 1), (gsum / max(n_upd, 1) if gsum is not None else torch.zeros((), device=device)).reshape(1)])
    if dist.is_initialized() and dist.get_world_size() > 1:  # one fused metric all-reduce per epoch
        dist.all_reduce(stats)
        stats /= dist.get_world_size()
    s = stats.tolist()
    return {"loss": s[0], "loss_learn": s[1], "loss_mse": s[2], "loss_chfr": s[3], "grad_norm": s[4], "lr": lr,
            # execution record (not in the reference's dict): clouds/s of this rank over the whole epoch and over the replayed part
            "clouds_per_s": seen / max(t_end - t0, 1e-9),
            "replay_clouds_per_s": seen_replay / t_replay if t_replay > 0 else 0.0,
            "replayed_iters": n_replay, "capture_s": t_capture}


def train_one_epoch(model, data_loader, optimizer, device, epoch, loss_scaler=None, log_writer=None, args=None,
                    model_ema=None, model_teacher=None, scheduler=None, optimizer_learn_loss=None,
                    grad_sync=None, print_freq=20, use_graph=None):
    """Drop-in for P/engine_pretrain.py::train_one_epoch (same positional arguments; `loss_scaler` is
    accepted for signature compatibility -- bf16/fp32 need no GradScaler).  The data loader yields this rank's
    (B,N,3) float batches (shard_for_rank gives the DistributedSampler partition the reference omits).  Returns the epoch's
    averaged stats like the reference (:268-271).

    Execution: with the flat optimizer (build_optimizer(flat=True, model_ema=...)) on a GPU the iterations are hipGraph
    replays -- the configuration bench.py measures: one graph on a single GPU; with a process group of >1 ranks the four-graph
    SegmentedDDPStep (optimizer built with segment_of=ddp_segment) or two graphs around GradSync.from_flat's all-reduce (created
    here when the caller passed none).  The capture happens once per epoch (the epoch's mask schedule and EMA decay are baked
    in), after the process's first EAGER_WARMUP_ITERS iterations, which run eagerly.  use_graph=False (or a torch optimizer, or
    GM3D_EAGER_EPOCH=1) keeps every iteration eager.  args.accum_iter follows P/:72-73,196-212: lr set, and clip / AdamW / EMA /
    zero_grad run, once per window."""
    import os
    assert args.learning_loss and model_ema is not None, "the north-star path trains with the EMA teacher"
    model.train(True)
    model_ema.decay = ema_decay_for_epoch(epoch)
    device = torch.device(device)
    flat = hasattr(optimizer, "flat_grad_views")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1 and grad_sync is None and flat:
        grad_sync = getattr(optimizer, "_grad_sync", None)
        if grad_sync is None:       # the all-reduce works in place on the optimizer's flat gradient buffer; also broadcasts rank 0's state
            grad_sync = optimizer._grad_sync = GradSync.from_flat(optimizer, bucket_bytes=256 << 20)
    if use_graph is None:
        use_graph = flat and device.type == "cuda" and os.environ.get("GM3D_EAGER_EPOCH") != "1"

    def eager_step(samples, it):
        return pretrain_step(model, model_ema, optimizer, samples, epoch, args, grad_sync=grad_sync, micro_step=it)

    def capture(example):
        return make_captured_step(model, model_ema, optimizer, args, example, epoch, grad_sync=grad_sync)

    return run_epoch(data_loader, optimizer, device, epoch, args, eager_step, capture, log_writer=log_writer,
                     print_freq=print_freq, model_key=model, use_graph=use_graph)
