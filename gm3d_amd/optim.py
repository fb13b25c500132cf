"""FlatAdamWEma: gradient clipping + AdamW + EMA teacher + bf16 GEMM shadows as ONE pass over flat buffers.

Mirrors, with identical arithmetic, what the reference does in three places -- NativeScalerWithGradNormCount.__call__
(P/util/misc.py:256-270: clip_grad_norm_(5.0) then optimizer.step()), the AdamW built by P/tools/builder.py:40-56 (no
weight decay for 1-D / bias / token parameters) and timm ModelEma.update (P/engine_pretrain.py:212) -- but lays the state
out for the machine: every parameter of the student is a view into one contiguous fp32 buffer (decayed parameters first),
and so are its gradient slot, both Adam moments, the EMA teacher's parameters and the bf16 copies the GEMMs read.
One step = gm3d_adamw_ema_flat_step (3 launches) instead of ~25 multi-tensor launches and ~9 passes over 147 MB.
"""
import os
import re
import weakref

import torch

from ._capi import check, lib
from .ops import _ptr, _stream


def _builder_no_decay(name, p):
    """P/tools/builder.py:40-56."""
    return len(p.shape) == 1 or name.endswith(".bias") or "token" in name


_BLOCK_RE = re.compile(r"^(.*\.blocks)\.(\d+)\.(.+)$")


def _kind_key(name):
    """Sort key that puts the same-kind weights of one block stack next to each other, in block order
    (X.blocks.0.mlp.fc1.weight, X.blocks.1.mlp.fc1.weight, ...): the stack's batched weight-gradient GEMM then has ONE
    contiguous (nblk, N, K) destination in the flat gradient buffer and writes it directly (fused._wgrad_batched)."""
    m = _BLOCK_RE.match(name)
    return (0, "", "", 0) if m is None else (1, m.group(1), m.group(3), int(m.group(2)))


def _split_decay(model, segment_of=None, no_decay_of=None):
    no_decay_of = no_decay_of or _builder_no_decay
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if no_decay_of(name, p) else decay).append((name, p))
    seg = segment_of or (lambda n: 0)
    # stable sorts: parameters outside the block stacks keep their order inside a segment
    decay.sort(key=lambda kv: (seg(kv[0]),) + _kind_key(kv[0]))
    if segment_of is not None:
        no_decay.sort(key=lambda kv: seg(kv[0]))
    return decay, no_decay


class _GradSlots:
    """parameter -> its slot (a view) in a flat gradient buffer.  A backward node that produces the gradients of several
    parameters in one tensor asks `stacked(params)` for a single destination covering their slots, when they are adjacent."""

    def __init__(self):
        self._slot = {}

    def register(self, p, view):
        self._slot[id(p)] = (weakref.ref(p), view)

    def get(self, p):
        hit = self._slot.get(id(p))
        return hit[1] if hit is not None and hit[0]() is p else None

    def stacked(self, params):
        """params: same-shape parameters -> (len(params), *shape) view of the flat buffer covering their slots, or None."""
        v0 = self.get(params[0])
        if v0 is None or not ENABLE_DIRECT_WGRAD:
            return None
        n, step = v0.numel(), v0.numel() * v0.element_size()
        for i, p in enumerate(params):
            v = self.get(p)
            if v is None or v.shape != v0.shape or v.data_ptr() != v0.data_ptr() + i * step:
                return None
        return torch.as_strided(v0, (len(params),) + tuple(v0.shape), (n,) + tuple(v0.stride()))


grad_slots = _GradSlots()
ENABLE_DIRECT_WGRAD = True    # weight gradients written straight into the flat buffer (tests flip it)


class FlatAdamWEma(torch.optim.Optimizer):
    def __init__(self, model, model_ema=None, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8, max_norm=5.0,
                 segment_of=None, no_decay_of=None, lr_scale_of=None):
        """no_decay_of(name, p) -> bool (optional) replaces the pretraining rule for which parameters skip weight decay;
        lr_scale_of(name) -> float (optional) gives every parameter a learning-rate multiplier (fine-tuning's layer-wise lr
        decay, P/util/lr_decay.py): the step then reads a per-element multiplier buffer laid out like the parameters.
        segment_of(name) -> int (optional): lay the decayed parameters out segment by segment (ascending), so that the
        gradients of one backward segment form ONE contiguous range of the flat buffer (`segment_ranges`: data-parallel runs
        all-reduce a segment's range while the next segment's backward is still running).  All non-decayed parameters (biases,
        LayerNorm / BatchNorm affine, tokens: <1 % of the bytes) follow the last segment and travel with it."""
        decay, no_decay = _split_decay(model, segment_of, no_decay_of)
        named = decay + no_decay
        params = [p for _, p in named]
        dev = params[0].device
        if not params[0].is_cuda:
            raise RuntimeError("FlatAdamWEma needs the model on the GPU (gm3d_amd has no CPU fallback)")
        pad = lambda k: (k + 3) // 4 * 4
        offs, off = [], 0
        for i, p in enumerate(params):
            if i == len(decay):
                off = pad(off)
                self.n_decay = off
            offs.append(off)
            off += p.numel()
        if not no_decay:
            self.n_decay = pad(off)
        self.n = pad(off)
        f32 = dict(dtype=torch.float32, device=dev)
        self.P, self.G = torch.zeros(self.n, **f32), torch.zeros(self.n, **f32)
        self.M, self.V = torch.zeros(self.n, **f32), torch.zeros(self.n, **f32)
        self.PS = torch.zeros(self.n, dtype=torch.bfloat16, device=dev)
        self.gviews = []
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.P[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                self.gviews.append(self.G[o:o + p.numel()].view_as(p))
                grad_slots.register(p, self.gviews[-1])
            self.PS.copy_(self.P)
        self.ema = model_ema
        self.model = model          # kept for the data-parallel start-up broadcast (engine_pretrain.broadcast_state)
        self.GA = None              # gradient-accumulation buffer (accum_iter > 1), allocated on first use
        self.E = self.ES = None
        if model_ema is not None:
            tparams = dict(model_ema.ema.named_parameters())
            self.E = torch.zeros(self.n, **f32)
            self.ES = torch.zeros(self.n, dtype=torch.bfloat16, device=dev)
            with torch.no_grad():
                for (name, p), o in zip(named, offs):
                    t = tparams[name]
                    view = self.E[o:o + p.numel()].view_as(t)
                    view.copy_(t)
                    t.data = view
                self.ES.copy_(self.E)
            model_ema.params_in_optimizer = True
            model_ema._pairs = None
            if hasattr(model_ema, "prepare"):
                # the teacher's buffer pairs (and the re-pointing of the BatchNorm counters into one tensor) are settled NOW: a step
                # captured later must not find them unbuilt (ModelEma.update refuses to build inside a capture)
                model_ema.prepare(model)
        self.LS = None
        if lr_scale_of is not None:
            self.LS = torch.ones(self.n, **f32)
            for (name, p), o in zip(named, offs):
                self.LS[o:o + p.numel()] = float(lr_scale_of(name))
        self.lr_dev = torch.tensor(float(lr), **f32)
        self.step_dev = torch.zeros(1, **f32)
        self.ema_w_dev = torch.zeros(1, **f32)
        self.scal = torch.zeros(4, **f32)
        self.partial = torch.zeros(max(lib.gm3d_flat_partial_rows(self.n), 1), **f32)
        self.max_norm = float(max_norm)
        defaults = dict(lr=self.lr_dev, weight_decay=weight_decay, betas=betas, eps=eps)
        super().__init__([{"params": params}], defaults)
        self.param_groups[0]["lr"] = self.lr_dev            # adjust_learning_rate fills this tensor in place
        for p, o in zip(params, offs):                       # AdamW-shaped state (views) so state_dict() looks familiar
            self.state[p] = {"step": self.step_dev, "exp_avg": self.M[o:o + p.numel()].view_as(p),
                             "exp_avg_sq": self.V[o:o + p.numel()].view_as(p)}
        self._params, self._offs, self._named = params, offs, named
        self.segment_ranges = None
        if segment_of is not None:
            segs = sorted({segment_of(n) for n, _ in decay})
            starts = {}
            for (n, _), o in zip(decay, offs):
                starts.setdefault(segment_of(n), o)
            bounds = [starts[sg] for sg in segs] + [self.n]
            self.segment_ranges = {sg: (bounds[i], bounds[i + 1] if i + 1 < len(segs) else self.n) for i, sg in enumerate(segs)}
            self.segment_params = {sg: [p for n, p in named if segment_of(n) == sg] for sg in segs}
        self._register_shadows(model, model_ema)

    def _register_shadows(self, model, model_ema):
        """GEMM code asks weight_cache for the bf16 copy of a weight: hand it views of the flat shadows, which the
        optimizer kernel rewrites every step (no separate cast launches)."""
        from .fused import weight_cache
        for p, o in zip(self._params, self._offs):
            weight_cache.pin_view(p, self.PS[o:o + p.numel()].view_as(p))
        if model_ema is not None:
            tparams = dict(model_ema.ema.named_parameters())
            for (name, p), o in zip(self._named, self._offs):
                weight_cache.pin_view(tparams[name], self.ES[o:o + p.numel()].view_as(p))

    def flat_grad_views(self):
        """The gradient slots: set `p.grad = view` to have backward accumulate straight into the flat buffer
        (data-parallel runs all-reduce chunks of `self.G`)."""
        return list(zip(self._params, self.gviews))

    def zero_grad(self, set_to_none=True):
        self._gathered = False          # whatever backward produces next has to be gathered again
        if set_to_none:
            for p in self._params:
                p.grad = None
        else:
            self.G.zero_()
            for p, g in zip(self._params, self.gviews):
                p.grad = g

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() plus `param_names`: the parameter name of every index, in this optimizer's own
        order (decayed parameters first, block-stack weights grouped by kind).  That order is NOT the reference AdamW's (two
        groups [no_decay, decay] in named_parameters order over all 470 tensors incl. the dead ones, P/tools/builder.py:40-56),
        so the 'optimizer' entry of a checkpoint is interchangeable between gm3d_amd runs only; model and EMA tensors are
        interchangeable with the reference (checkpoint.py)."""
        sd = super().state_dict()
        sd["param_names"] = [n for n, _ in self._named]
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Restore from `state_dict()` of a FlatAdamWEma over the same model: the moments are copied INTO the flat buffers
        (torch's default would re-point the state at fresh tensors and break the layout).  Entries are matched by parameter
        NAME when the file carries `param_names` (any layout order), by position otherwise; shapes are verified either way."""
        ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        if len(ids) != len(self._params):
            raise ValueError("optimizer state has %d parameters, this model %d" % (len(ids), len(self._params)))
        names = state_dict.get("param_names")
        if names is not None:
            if sorted(names) != sorted(n for n, _ in self._named):
                raise ValueError("optimizer state was written for a different set of parameters")
            by_name = dict(zip(names, ids))
            ids = [by_name[n] for n, _ in self._named]
        state = state_dict["state"]
        step = 0.0
        for pid, p, o, (name, _) in zip(ids, self._params, self._offs, self._named):
            st = state.get(pid, state.get(str(pid)))
            if st is None:
                continue
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape) or tuple(st["exp_avg_sq"].shape) != tuple(p.shape):
                raise ValueError("optimizer state of %s has shape %s, the parameter %s"
                                 % (name, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.M[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.V[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, float(st["step"]))
        self.step_dev.fill_(step)
        g0 = state_dict["param_groups"][0]
        self.lr_dev.fill_(float(g0["lr"]))
        for k in ("weight_decay", "betas", "eps"):
            if k in g0:
                self.param_groups[0][k] = tuple(g0[k]) if k == "betas" else g0[k]

    @torch.no_grad()
    def sync_shadows(self):
        """After parameters were written from outside (checkpoint load): refresh the bf16 GEMM copies."""
        self.PS.copy_(self.P)
        if self.E is not None:
            self.ES.copy_(self.E)

    def mark_grads_filled(self):
        """The caller wrote every gradient into `G` itself (the segmented data-parallel step stores each backward segment into
        its range): the next step() must not gather from p.grad."""
        self._gathered = True

    @torch.no_grad()
    def accumulate(self, last):
        """Gradient accumulation over micro-batches (accum_iter > 1; P/engine_pretrain.py:195-212 accumulates into p.grad): call
        after gather_grads() of every micro-batch.  G holds this micro-batch's gradients (backward nodes overwrite their slots),
        GA the running sum: not last -> GA += G; last -> G += GA, GA = 0, so that clip / all-reduce / AdamW see the sum."""
        if self.GA is None:
            self.GA = torch.zeros_like(self.G)
        if last:
            self.G.add_(self.GA)
            self.GA.zero_()
        else:
            self.GA.add_(self.G)
            self._gathered = False      # the next micro-batch gathers again

    @torch.no_grad()
    def gather_grads(self):
        """Copy the gradients autograd produced (fresh tensors, no accumulate kernels) into the flat buffer with one
        multi-tensor launch; data-parallel runs call this at the end of backward so the all-reduce can work on `G`.
        Slots of parameters that received no gradient this iteration are zeroed (never left at the previous step's values)."""
        if getattr(self, "_gathered", False):       # mark_grads_filled(): G is complete already
            return
        dst, src, zero = [], [], []
        for p, v in zip(self._params, self.gviews):
            g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr():      # else: the backward node wrote this gradient into its slot already
                dst.append(v)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        if zero:
            torch._foreach_zero_(zero)
        self._gathered = True

    @torch.no_grad()
    def step(self, closure=None):
        """-> gradient norm before clipping (device scalar)."""
        if not getattr(self, "_gathered", False):
            self.gather_grads()
        self._gathered = False
        g = self.param_groups[0]
        if self.ema is not None:
            # the EMA weight on the device: eager steps write it only when it changes (once per epoch).  A captured step always
            # carries its own fill -- a graph must not depend on what an earlier eager step left on the device -- and forgets the
            # host-side record, since its replays change the device value behind the host's back (a later eager step
            # at the recorded decay would otherwise skip the fill and run with the graph's weight)
            w = 1.0 - float(self.ema.decay)
            if self.ema_w_dev.is_cuda and torch.cuda.is_current_stream_capturing():
                self.ema_w_dev.fill_(w)
                self._ema_w_host = None
            elif getattr(self, "_ema_w_host", None) != w:
                self.ema_w_dev.fill_(w)
                self._ema_w_host = w
        from .ops import _launch
        _launch("gm3d_adamw_ema_flat_step", {"n": self.n, "ema": self.E is not None}, lib.gm3d_adamw_ema_flat_step_lrd,
                _ptr(self.P), _ptr(self.G), _ptr(self.M), _ptr(self.V), _ptr(self.E), _ptr(self.PS), _ptr(self.ES), self.n,
                self.n_decay, _ptr(self.lr_dev), _ptr(self.LS), float(g["weight_decay"]), float(g["betas"][0]),
                float(g["betas"][1]), float(g["eps"]), _ptr(self.ema_w_dev), self.max_norm, _ptr(self.step_dev),
                _ptr(self.partial), _ptr(self.scal), _stream())
        return self.scal[3]
