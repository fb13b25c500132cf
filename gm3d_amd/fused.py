"""Fused transformer stack: the blocks of TransformerEncoder / TransformerDecoder plus their final LayerNorm
as ONE autograd node with a hand-written forward and backward.

Restates, operation for operation, timm-0.4.5 Block (in-tree twin Point-MAE_SA3D/models/Point_MAE.py:82-146)
as driven by P/models_mae_learn_loss.py:914-917 and :984-990:

    for block in blocks:  x = block(x + pos)        # x += dp(attn(LN1(x)));  x += dp(mlp(LN2(x)))
    return LayerNorm(x)

Per block the forward is 4 GEMMs (hipBLASLt via torch.mm on cached bf16 weights), the MFMA attention kernel and
3 row-wise HIP passes (gm3d_amd/csrc/rowops.hip): [residual + bias + DropPath + pos + LayerNorm] x2 and
[bias + GELU]; the backward is 8 GEMMs, the attention backward and 3 row-wise passes that also produce every
bias / gamma / beta gradient as fused column sums.  The residual stream is fp32, GEMM operands are bf16
(throughput mode) or fp32 (parity mode: same code path, checked to 1e-5 against the reference fixtures).
"""
import weakref

import contextlib

import torch

from ._capi import lib
from . import gemm
from . import streams as _streams
from .ops import _launch, _ptr, _stream, _DT

LNC = 384


# ----------------------------------------------------------------------------- bf16 weight cache
class _WeightCache:
    """bf16 copies of the 2-D fp32 master weights for the GEMMs.

    * `pin(model)` / `refresh()`: persistent bf16 buffers for every >=2-D parameter of a model, re-cast by ONE
      multi-tensor launch (the engine calls refresh() once per step for student and teacher; autocast re-casts
      ~400 times per step).  Under hipGraph capture this is the only mode that is correct: the refresh launch is
      part of the captured step, and get() involves no host-side staleness decision.
    * un-pinned weights: cached per (storage, version counter); re-cast when the master changed; never cached
      while a stream capture is running."""

    def __init__(self):
        self._c = {}
        self._pinned = {}      # data_ptr -> bf16 buffer
        self._groups = []      # (fp32 list, bf16 list)

    def pin(self, model, dtype=torch.bfloat16, static=False):
        """static=True: frozen weights -- cast once, never re-cast by refresh()."""
        src = [p for p in model.parameters() if p.dim() >= 2 and p.dtype == torch.float32 and p.is_cuda
               and not self._is_pinned(p)]
        if not src:
            return
        dst = [torch.empty_like(p, dtype=dtype) for p in src]
        for p, d in zip(src, dst):
            self._pinned[p.data_ptr()] = (d, weakref.ref(p))
        if static:
            with torch.no_grad():
                torch._foreach_copy_(dst, [p.detach() for p in src])
            return
        self._groups.append(([p.detach() for p in src], dst))
        with torch.no_grad():
            torch._foreach_copy_(dst, self._groups[-1][0])

    def pin_view(self, p, shadow):
        """Register an externally maintained bf16 copy of `p` (FlatAdamWEma rewrites it inside its update kernel)."""
        self._pinned[p.data_ptr()] = (shadow, weakref.ref(p))
        gemm.forget_transposes()        # shadows of another optimizer: whatever mm_nn remembered is stale

    def _is_pinned(self, p):
        hit = self._pinned.get(p.data_ptr())
        return hit is not None and hit[1]() is p

    def refresh(self):
        with torch.no_grad():
            for src, dst in self._groups:
                torch._foreach_copy_(dst, src)

    def get(self, w, dtype):
        if w.dtype == dtype:
            return w
        hit = self._pinned.get(w.data_ptr())
        if hit is not None and hit[1]() is w and hit[0].dtype == dtype:
            return hit[0]
        base = w._base
        if base is not None and base.dtype == w.dtype:
            # a view of a pinned parameter (the [global | local] column halves of a mini-PointNet conv): the same view of its shadow
            hit = self._pinned.get(base.data_ptr())
            if hit is not None and hit[1]() is base and hit[0].dtype == dtype and hit[0].numel() == base.numel():
                return torch.as_strided(hit[0], w.size(), w.stride(), hit[0].storage_offset() + w.storage_offset() - base.storage_offset())
        if torch.cuda.is_current_stream_capturing():
            return w.detach().to(dtype)
        key = (w.data_ptr(), dtype, tuple(w.shape))
        hit = self._c.get(key)
        if hit is not None and hit[0] == w._version and hit[2]() is w:   # same live tensor, unchanged since the cast
            if w.is_cuda:
                # the cast may have been launched on ANOTHER stream a moment ago (the parallel inference chains of run_stack meet the
                # same weights one after the other): whoever reuses it waits for the casting launch and keeps the allocator informed
                cur = torch.cuda.current_stream(w.device)
                if hit[3] != cur.cuda_stream:
                    cur.wait_event(hit[4])
                    hit[1].record_stream(cur)
            return hit[1]
        with torch.no_grad():
            c = w.detach().to(dtype)
        if len(self._c) > 4096:
            self._c.clear()
        stream_id, done = None, None
        if w.is_cuda:
            cur = torch.cuda.current_stream(w.device)
            stream_id, done = cur.cuda_stream, torch.cuda.Event()
            done.record(cur)
        self._c[key] = (w._version, c, weakref.ref(w), stream_id, done)
        return c


weight_cache = _WeightCache()


# ----------------------------------------------------------------------------- row-op wrappers
def residual_ln_fwd(res, y, bias, rowscale, rows_per_sample, add, gamma, beta, eps, adt, R, want_res=True, h=None):
    dev = gamma.device
    out_res = torch.empty(R, LNC, dtype=torch.float32, device=dev) if want_res else None
    if h is None:
        h = torch.empty(R, LNC, dtype=adt, device=dev)
    mean = torch.empty(R, dtype=torch.float32, device=dev)
    rstd = torch.empty(R, dtype=torch.float32, device=dev)
    _launch("gm3d_residual_ln_fwd", {"R": R, "dtype": str(adt)}, lib.gm3d_residual_ln_fwd, _ptr(res), _ptr(y), _ptr(bias),
            _ptr(rowscale), int(rows_per_sample), _ptr(add), _ptr(gamma), _ptr(beta), float(eps), _ptr(out_res), _ptr(h),
            _ptr(mean), _ptr(rstd), R, LNC, _DT[adt], _stream())
    return out_res, h, mean, rstd


def residual_ln_bwd(dh, gin, x, mean, rstd, gamma, rowscale, rows_per_sample, acc, want_dy, adt, R, dy=None, partial=None,
                    acc_out=None):
    """-> dx (R,C) f32, dy (R,C) adt | None, sums (3,C) f32 = [dgamma, dbeta, colsum(dy)].
    With `partial` given (a (rows, 3C) slice of a batched buffer) the second stage is left to the caller.
    acc_out (R,C) adt: receives the updated `acc` in the activation type too."""
    dev = gamma.device
    dx = torch.empty(R, LNC, dtype=torch.float32, device=dev)
    if dy is None and want_dy:
        dy = torch.empty(R, LNC, dtype=adt, device=dev)
    nrows = lib.gm3d_ln_partial_rows(R)
    defer = partial is not None
    if partial is None:
        partial = torch.empty(nrows, 3 * LNC, dtype=torch.float32, device=dev)
    _launch("gm3d_residual_ln_bwd", {"R": R, "dtype": str(adt)}, lib.gm3d_residual_ln_bwd, _ptr(dh), _ptr(gin), _ptr(x),
            _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(rowscale), int(rows_per_sample), _ptr(dx), _ptr(dy), _ptr(acc),
            _ptr(acc_out), _ptr(partial), R, LNC, _DT[adt], _stream())
    if defer:
        return dx, dy, None
    sums = torch.empty(3, LNC, dtype=torch.float32, device=dev)
    _launch("gm3d_colsum_finish", {"rows": nrows, "cols": 3 * LNC}, lib.gm3d_colsum_finish, _ptr(partial), nrows, 3 * LNC,
            3 * LNC, _ptr(sums), 0, _stream())
    return dx, dy, sums


def bias_gelu_fwd(f, bias, adt, g=None):
    R, C = f.shape
    if g is None:
        g = torch.empty_like(f)
    _launch("gm3d_bias_gelu_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_bias_gelu_fwd, _ptr(f), _ptr(bias), _ptr(g),
            R, C, _DT[adt], _stream())
    return g


def bias_gelu_bwd(dg, f, bias, adt, df=None, partial=None):
    """-> df (R,C) adt, dbias (C) f32 (None when `partial`, a slice of a batched buffer, is given)."""
    R, C = f.shape
    if df is None:
        df = torch.empty_like(f)
    nrows = lib.gm3d_gelu_partial_rows(R)
    defer = partial is not None
    if partial is None:
        partial = torch.empty(nrows, C, dtype=torch.float32, device=f.device)
    _launch("gm3d_bias_gelu_bwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_bias_gelu_bwd, _ptr(dg), _ptr(f), _ptr(bias),
            _ptr(df), _ptr(partial), R, C, _DT[adt], _stream())
    if defer:
        return df, None
    db = torch.empty(C, dtype=torch.float32, device=f.device)
    _launch("gm3d_colsum_finish", {"rows": nrows, "cols": C}, lib.gm3d_colsum_finish, _ptr(partial), nrows, C, C, _ptr(db), 0,
            _stream())
    return df, db


def finish_batched(partial, out):
    """partial (J, rows, cols) f32 -> out (J, cols): every job's second stage in one launch."""
    J, rows, cols = partial.shape
    _launch("gm3d_colsum_finish_batched", {"rows": J * rows, "cols": cols}, lib.gm3d_colsum_finish_batched, _ptr(partial), J,
            rows * cols, rows, cols, cols, _ptr(out), cols, _stream())
    return out


def _attention_fwd(qkv, B, T, H, scale, out=None):
    if out is None:
        out = torch.empty(B * T, H * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    if T > 128:      # more tokens than one workgroup's LDS images hold (cfgs/config_3.yaml: 256 groups): the flash-style kernel of
        #              the hierarchical encoder with no mask (T <= 512; same qkv / out / lse layouts)
        _launch("gm3d_attention_masked_fwd", {"B": B, "T": T, "H": H, "HD": 64, "dtype": str(qkv.dtype)}, lib.gm3d_attention_masked_fwd,
                _ptr(qkv), None, _ptr(out), _ptr(lse), B, T, H, 64, float(scale), _DT[qkv.dtype], _stream())
        return out, lse
    _launch("gm3d_attention_fwd", {"B": B, "T": T, "H": H, "dtype": str(qkv.dtype)}, lib.gm3d_attention_fwd, _ptr(qkv),
            _ptr(out), _ptr(lse), B, T, H, float(scale), _DT[qkv.dtype], _stream())
    return out, lse


def _attention_qkv_fwd(h, wqkv, B, T, H, scale, out=None, want_qkv=False, want_lse=False):
    """qkv projection + attention in one launch (csrc/attention.hip attn_qkv_fwd_bf16_kernel); returns (out, lse|None, qkv|None)."""
    C = h.shape[1]
    if out is None:
        out = torch.empty(B * T, C, dtype=h.dtype, device=h.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=h.device) if want_lse else None
    qkv = torch.empty(B * T, 3 * C, dtype=h.dtype, device=h.device) if want_qkv else None
    _launch("gm3d_attention_qkv_fwd", {"B": B, "T": T, "H": H, "C": C, "dtype": str(h.dtype)}, lib.gm3d_attention_qkv_fwd, _ptr(h),
            _ptr(wqkv), _ptr(out), _ptr(lse) if want_lse else None, _ptr(qkv) if want_qkv else None, B, T, H, C, float(scale),
            _DT[h.dtype], _stream())
    return out, lse, qkv


def attention_qkv_supported(h, wqkv, T, H):
    return (h.is_cuda and h.dtype == torch.bfloat16 and wqkv.dtype == torch.bfloat16 and T <= 64 and h.shape[1] == 384 == H * 64
            and h.is_contiguous() and wqkv.is_contiguous() and tuple(wqkv.shape) == (1152, 384)
            and h.data_ptr() % 16 == 0 and wqkv.data_ptr() % 16 == 0)


def _attention_bwd(qkv, out, dout, lse, B, T, H, scale, dqkv=None):
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if T > 128:
        _launch("gm3d_attention_masked_bwd", {"B": B, "T": T, "H": H, "HD": 64, "dtype": str(qkv.dtype)}, lib.gm3d_attention_masked_bwd,
                _ptr(qkv), None, _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, T, H, 64, float(scale), _DT[qkv.dtype], _stream())
        return dqkv
    _launch("gm3d_attention_bwd", {"B": B, "T": T, "H": H, "dtype": str(qkv.dtype)}, lib.gm3d_attention_bwd, _ptr(qkv),
            _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, T, H, float(scale), _DT[qkv.dtype], _stream())
    return dqkv


def _wgrad_batched(dy, x, out=None):
    """dW[i] = dy[i]^T @ x[i] for every block of a stack in ONE batched GEMM, fp32 result.
    out (nblk,N,K) f32 (optional): the destination -- the parameters' slots in the optimizer's flat gradient buffer.
    (K = rows is 3200..8192 and the output is only 384..1536 wide: a single such GEMM fills a fraction of the
    256 CUs -- 50 us each through hipBLASLt -- while the batch over blocks runs at ~16 us per block.)"""
    nblk, R, N = dy.shape
    K = x.shape[2]
    if gemm.wgrad_supported(dy, x) and (out is None or (out.stride(2) == 1 and out.stride(1) == K and out.stride(0) == N * K)):
        return gemm.wgrad_nt(dy, x, out)       # hand-written NT kernel: LDS-DMA staging + transposed LDS reads (csrc/gemm_nt.hip)
    S = WGRAD_ROW_SPLIT
    if S == 0:      # auto: the smallest split that gives the chip >= 512 output tiles of 128x128 while keeping >= 512 rows each
        S, tiles = 1, nblk * max(N // 128, 1) * max(K // 128, 1)
        while tiles * S < 512 and S < 8 and R % (2 * S) == 0 and R // (2 * S) >= 512:
            S *= 2
    elif nblk > 4 or R // S < 1024:
        S = 1
    if S > 1 and R % S == 0 and dy.is_contiguous() and x.is_contiguous():
        # few blocks (the 4-block decoders): nblk * (N/128) * (K/128) tiles do not fill 256 CUs and each runs an 8192-deep
        # reduction -- split the rows S ways into more batches and add the S partial products with the two-stage sum
        part = _wgrad_batched_plain(dy.view(nblk * S, R // S, N), x.view(nblk * S, R // S, K))
        if out is None:
            out = torch.empty(nblk, N, K, dtype=torch.float32, device=dy.device)
        if SUM_FEW_ROWS:
            _launch("gm3d_sum_few_rows", {"rows": nblk * S, "cols": N * K}, lib.gm3d_sum_few_rows, _ptr(part), nblk, S, N * K,
                    _ptr(out), _stream())
        else:
            _launch("gm3d_colsum_finish_batched", {"rows": nblk * S, "cols": N * K}, lib.gm3d_colsum_finish_batched, _ptr(part), nblk,
                    S * N * K, S, N * K, N * K, _ptr(out), N * K, _stream())
        return out
    return _wgrad_batched_plain(dy, x, out)


SUM_FEW_ROWS = True
WGRAD_ROW_SPLIT = 4      # library fallback only (shapes the NT kernel does not take): 0 = auto, 1 = off, n = n-way for <=4-block stacks


_BMM_OUT_OK = [True]     # torch.bmm(..., out_dtype=, out=) available?


def _wgrad_batched_plain(dy, x, out=None):
    if dy.dtype == torch.float32:
        return torch.bmm(dy.transpose(1, 2), x) if out is None else torch.bmm(dy.transpose(1, 2), x, out=out)
    if out is not None and _BMM_OUT_OK[0]:
        try:
            return torch.bmm(dy.transpose(1, 2), x, out_dtype=torch.float32, out=out)
        except (TypeError, RuntimeError):
            _BMM_OUT_OK[0] = False
    try:
        r = torch.bmm(dy.transpose(1, 2), x, out_dtype=torch.float32)
    except TypeError:
        r = torch.bmm(dy.transpose(1, 2), x).float()
    return r if out is None else out.copy_(r)


WGRAD_MIN_BLOCKS = 5      # stacks with fewer blocks (the 4-block decoders) DEFER their weight-gradient GEMMs to the deep stack's fork point
_ASYNC_WGRAD = {"stream": None, "used": False, "min_blocks": 5, "deferred": []}


def _wgrad_many(reqs):
    """[(dy, x, out slot | None), ...] -> [dW, ...]: one launch for all of them where the NT kernel takes every request
    (gemm.wgrad_nt_multi), else one batched GEMM per request."""
    if gemm.MULTI_WGRAD and len(reqs) > 1 and all(gemm.wgrad_multi_ok(dy, x, out) for dy, x, out in reqs):
        if len(reqs) <= 16:
            return gemm.wgrad_nt_multi(reqs)
        res = []                      # the kernel takes 16 problems per launch: ceil(n / 16) launches of about equal size
        per = -(-len(reqs) // -(-len(reqs) // 16))
        for i in range(0, len(reqs), per):
            chunk = reqs[i:i + per]
            res += gemm.wgrad_nt_multi(chunk) if len(chunk) > 1 else [_wgrad_batched(*chunk[0])]
        return res
    return [_wgrad_batched(dy, x, out) for dy, x, out in reqs]


def defer_wgrad(dy, x, w=None):
    """dW (N,K) f32 = dy (R,N)^T @ x (R,K) for a layer OUTSIDE the block stacks (heads, positional MLP), w its weight parameter.
    Inside the engine's backward region, when w has a slot in the optimizer's flat gradient buffer and no gradient yet (nothing will
    read or add to the result before the region ends), the product joins the stacks' one-launch weight gradients and lands in the slot;
    elsewhere, or when the kernel does not take the shape, it is computed now."""
    reg = _ASYNC_WGRAD
    if (reg["stream"] is not None and gemm.MULTI_WGRAD and (DEFER_HEAD_WGRAD or reg.get("defer_heads")) and w is not None and w.is_leaf and w.grad is None
            and dy.dim() == 2
            and dy.is_contiguous() and x.is_contiguous()):
        from .optim import grad_slots, ENABLE_DIRECT_WGRAD
        slot = grad_slots.get(w) if ENABLE_DIRECT_WGRAD else None
        if slot is not None and slot.is_contiguous() and slot.numel() == dy.shape[1] * x.shape[1]:
            out = slot.view(1, dy.shape[1], x.shape[1])
            if gemm.wgrad_multi_ok(dy.unsqueeze(0), x.unsqueeze(0), out):
                reg["deferred"].append(((dy, x), [(dy.unsqueeze(0), x.unsqueeze(0), out)]))
                return out[0]
    from .embed import splitk_wgrad
    return splitk_wgrad(dy, x)


DEFER_HEAD_WGRAD = False  # the heads' / positional MLP's weight gradients inside the stacks' one-launch form (fused.defer_wgrad): measured
#                           0.4 % slower (7.43 vs 7.39 ms same-box): three small products are cheaper where they stand than as extra tiles there


def _run_deferred(reg, own=()):
    """the deferred stacks' weight-gradient requests, plus `own` (the calling stack's), on the current stream -> the results of `own`"""
    jobs, reg["deferred"] = reg["deferred"], []
    cur = torch.cuda.current_stream()
    reqs = []
    for tensors, rq in jobs:
        for t in tensors:
            t.record_stream(cur)
        reqs += rq
    res = _wgrad_many(reqs + list(own)) if (reqs or own) else []
    return res[len(reqs):]

_wgrad_streams = {}
ASYNC_WGRAD = True       # (tests flip it)


class async_wgrad:
    """with async_wgrad(device): <backward>  -- weight-gradient GEMMs of the deep block stacks run on a side stream during the
    region; on exit the current stream waits for it (so whatever follows -- gradient gather, all-reduce, optimizer -- sees them)."""

    def __init__(self, device, min_blocks=None, defer_heads=False):
        """defer_heads: weight gradients of layers outside the block stacks (fused.defer_wgrad: token embeds, heads, positional MLPs)
        join ONE launch at the region's exit -- for a model with many small such layers (Point-M2AE: 19) that is 2 launches instead
        of 38; for the north-star model's three it measured slower (DEFER_HEAD_WGRAD)."""
        self.dev, self.min_blocks = torch.device(device), (WGRAD_MIN_BLOCKS if min_blocks is None else min_blocks)
        self.defer_heads = bool(defer_heads)

    def __enter__(self):
        if ASYNC_WGRAD and self.dev.type == "cuda":
            key = (self.dev.type, self.dev.index)
            if key not in _wgrad_streams:
                _wgrad_streams[key] = torch.cuda.Stream(device=self.dev)
            _ASYNC_WGRAD.update(stream=_wgrad_streams[key], used=False, min_blocks=self.min_blocks, defer_heads=self.defer_heads)
            if self.defer_heads:
                gemm.prepare_transposes()       # the small layers' transposed weight shadows: one launch per eight, not one per layer
        return self

    def __exit__(self, *exc):
        reg = _ASYNC_WGRAD
        ws, used = reg["stream"], reg["used"]
        if reg.get("defer_heads"):
            gemm.drop_transposes()
        reg.update(stream=None, used=False, defer_heads=False)
        if reg["deferred"]:                 # no deep stack came by to take them along: run them here, in line
            if exc[0] is None:
                _run_deferred(reg)
            reg["deferred"] = []
        if ws is not None and used:
            _streams.join(ws)
        return False


DIRECT_INPUT_GRADS = True   # the stack backward writes dx / dpos in the activation type itself (no converting copies)
FUSE_QKV_ATTENTION = True   # qkv GEMM + attention forward as one launch (bf16, 32 < T <= 64)
PER_BLOCK = 11  # ln1.w ln1.b qkv.w proj.w proj.b ln2.w ln2.b fc1.w fc1.b fc2.w fc2.b


def block_params(block):
    return [block.norm1.weight, block.norm1.bias, block.attn.qkv.weight, block.attn.proj.weight, block.attn.proj.bias,
            block.norm2.weight, block.norm2.bias, block.mlp.fc1.weight, block.mlp.fc1.bias, block.mlp.fc2.weight,
            block.mlp.fc2.bias]


def _mm(x, w):
    """x (R,K) @ w (N,K)^T on the kernel gemm.choose names for the shape (hand-written register-prefetch / LDS-DMA ring / library)."""
    return gemm.mm(x, w)


class TransformerStackFn(torch.autograd.Function):
    """args: x (B,T,C), pos (B,T,C), meta, final_w, final_b, then PER_BLOCK tensors per block.
    meta: dict(num_heads, scale, eps, final_eps, adt, dp=[(scale_attn|None, scale_mlp|None) per block])."""

    @staticmethod
    def forward(ctx, x, pos, meta, final_w, final_b, *params):
        with torch.autocast("cuda", enabled=False):
            return TransformerStackFn._forward(ctx, x, pos, meta, final_w, final_b, *params)

    @staticmethod
    def _forward(ctx, x, pos, meta, final_w, final_b, *params):
        g = TransformerStackFn._forward_gen(ctx, x, pos, meta, final_w, final_b, *params)
        while True:
            try:
                next(g)
            except StopIteration as done:
                return done.value

    @staticmethod
    def _forward_gen(ctx, x, pos, meta, final_w, final_b, *params):
        """The forward as a generator that yields after every kernel launch: run_stack drives several half-batch chains in
        lock-step, one launch of each in turn on its own stream, so that a captured graph holds them as interleaved branches."""
        B, T, C = x.shape
        assert C == LNC and len(params) % PER_BLOCK == 0
        nblk = len(params) // PER_BLOCK
        adt, H, scale, eps = meta["adt"], meta["num_heads"], meta["scale"], meta["eps"]
        R = B * T
        posa = pos.reshape(R, C).to(adt).contiguous()
        need = meta["grad"] and any(ctx.needs_input_grad)   # grad mode is off inside forward: captured by run_stack
        y = bias = rs = None
        if x.dtype == adt and adt != torch.float32:
            # tokens already in the activation dtype: the first LayerNorm pass reads them as its branch operand (u = x + pos) --
            # no fp32 copy of the input
            res, y = None, x.reshape(R, C).contiguous()
        else:
            res = x.reshape(R, C).float().contiguous()
        saved = []
        dev = x.device
        if need:  # operands of the weight-gradient GEMMs, stacked over blocks for the batched wgrad
            H1 = torch.empty(nblk, R, C, dtype=adt, device=dev)
            A = torch.empty(nblk, R, C, dtype=adt, device=dev)
            H2 = torch.empty(nblk, R, C, dtype=adt, device=dev)
            GG = torch.empty(nblk, R, 4 * C, dtype=adt, device=dev)
        # LayerNorm folded into the GEMMs around it (bf16 path): proj / fc2 write the fp32 residual stream + per-tile row
        # statistics in their epilogue (gm3d_gemm_tn_bf16_res), qkv / fc1 normalise while they stage their A operand
        # (gm3d_gemm_tn_bf16_lna).  Stand-alone LayerNorm passes remain only in front of the first block and behind the last.
        fuse_ln = (gemm.ENABLED and gemm.FUSE_LN and gemm.FUSE_GELU and adt == torch.bfloat16 and x.is_cuda
                   and all(gemm.supported(posa, weight_cache.get(params[i * PER_BLOCK + k], adt)) or k == 9
                           for i in range(nblk) for k in (2, 3, 7)))
        st1 = None
        for i in range(nblk):
            ln1w, ln1b, wqkv, wproj, bproj, ln2w, ln2b, w1, b1, w2, b2 = params[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            dp1, dp2 = meta["dp"][i]
            if fuse_ln:
                if i == 0:
                    u, h1, m1, r1 = residual_ln_fwd(res, y, bias, rs, T, posa, ln1w, ln1b, eps, adt, R, h=H1[i] if need else None)
                    yield
                    qkv = _mm(h1, weight_cache.get(wqkv, adt))
                else:       # u, st1: the previous block's fc2 epilogue
                    qkv, m1, r1 = gemm.linear_lna(u16, st1, ln1w, ln1b, eps, weight_cache.get(wqkv, adt), None,
                                                  h_out=H1[i] if need else None, want_stats=need)
                yield
                a, lse = _attention_fwd(qkv, B, T, H, scale, out=A[i] if need else None)
                yield
                x1, x16, st2 = gemm.linear_res(a, weight_cache.get(wproj, adt), bproj, u, dp1, T, None)
                yield
                W1 = weight_cache.get(w1, adt)
                f, g, m2, r2 = gemm.linear_lna(x16, st2, ln2w, ln2b, eps, W1, b1, gelu=True,
                                               f_out=torch.empty(R, W1.shape[0], dtype=adt, device=dev) if need else None,
                                               g_out=GG[i] if need else None, h_out=H2[i] if need else None, want_stats=need)
                yield
                if need:
                    saved += [u, m1, r1, qkv, lse, x1, m2, r2, f]
                if i + 1 < nblk:
                    u, u16, st1 = gemm.linear_res(g, weight_cache.get(w2, adt), b2, x1, dp2, T, posa)
                else:
                    o = _mm(g, weight_cache.get(w2, adt))
                    res, y, bias, rs = x1, o, b2, dp2
                yield
                continue
            u, h1, m1, r1 = residual_ln_fwd(res, y, bias, rs, T, posa, ln1w, ln1b, eps, adt, R, h=H1[i] if need else None)
            yield
            Wq = weight_cache.get(wqkv, adt)
            if FUSE_QKV_ATTENTION and T > 32 and attention_qkv_supported(h1, Wq, T, H):
                # q|k|v never leave the CU in the no-grad (teacher) pass; with a backward to come they are written on the side
                a, lse, qkv = _attention_qkv_fwd(h1, Wq, B, T, H, scale, out=A[i] if need else None, want_qkv=need, want_lse=need)
            else:
                qkv = _mm(h1, Wq)
                yield
                a, lse = _attention_fwd(qkv, B, T, H, scale, out=A[i] if need else None)
            yield
            p = _mm(a, weight_cache.get(wproj, adt))
            yield
            x1, h2, m2, r2 = residual_ln_fwd(u, p, bproj, dp1, T, None, ln2w, ln2b, eps, adt, R, h=H2[i] if need else None)
            yield
            W1 = weight_cache.get(w1, adt)
            if gemm.FUSE_GELU and gemm.supported(h2, W1):
                # fc1 + bias + GELU in the GEMM epilogue; the pre-activation is only written when a backward follows
                fo = torch.empty(R, W1.shape[0], dtype=adt, device=dev) if need else None
                if gemm.dma_supported(h2, W1):
                    f, g = gemm.linear_gelu_dma(h2, W1, b1, f_out=fo, g_out=GG[i] if need else None, bm=gemm.dma_bm(R))
                else:
                    f, g = gemm.linear_gelu(h2, W1, b1, f_out=fo, g_out=GG[i] if need else None)
            else:
                f = h2 @ W1.t()
                yield
                g = bias_gelu_fwd(f, b1, adt, g=GG[i] if need else None)
            yield
            o = _mm(g, weight_cache.get(w2, adt))
            yield
            if need:
                saved += [u, m1, r1, qkv, lse, x1, m2, r2, f]
            res, y, bias, rs = x1, o, b2, dp2
        xf, hout, mf, rf = residual_ln_fwd(res, y, bias, rs, T, None, final_w, final_b, meta["final_eps"], adt, R,
                                           want_res=need, h=meta.get("hout"))
        if need:
            ctx.save_for_backward(final_w, xf, mf, rf, H1, A, H2, GG, *params, *saved)
        ctx.meta, ctx.shape, ctx.nblk = meta, (B, T, C), nblk
        ctx.in_dtypes = (x.dtype, pos.dtype)
        return hout.view(B, T, C)

    @staticmethod
    def backward(ctx, dout):
        with torch.autocast("cuda", enabled=False):
            return TransformerStackFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        meta, (B, T, C), nblk = ctx.meta, ctx.shape, ctx.nblk
        adt, H, scale = meta["adt"], meta["num_heads"], meta["scale"]
        R = B * T
        tens = ctx.saved_tensors
        final_w, xf, mf, rf, H1, A, H2, GG = tens[:8]
        params = tens[8:8 + nblk * PER_BLOCK]
        saved = tens[8 + nblk * PER_BLOCK:]
        grads = [None] * (nblk * PER_BLOCK)
        dev = dout.device
        dh = dout.reshape(R, C).to(adt).contiguous()
        # output-side operands of the weight-gradient GEMMs, stacked over blocks
        DO = torch.empty(nblk, R, C, dtype=adt, device=dev)
        DF = torch.empty(nblk, R, 4 * C, dtype=adt, device=dev)
        DP = torch.empty(nblk, R, C, dtype=adt, device=dev)
        DQ = torch.empty(nblk, R, 3 * C, dtype=adt, device=dev)
        # column-sum partials of every LayerNorm site (job 2i: LN1 of block i, 2i+1: LN2, 2*nblk: final) and of every
        # GELU site, finished by ONE launch each after the loop
        PLN = torch.empty(2 * nblk + 1, lib.gm3d_ln_partial_rows(R), 3 * C, dtype=torch.float32, device=dev)
        SLN = torch.empty(2 * nblk + 1, 3, C, dtype=torch.float32, device=dev)
        # fc2 input gradient + GELU backward + fc1 bias-gradient partials as ONE launch of the hand-written GEMM (bf16 path):
        # it wants fc2.weight^T (and proj.weight^T for the attention branch), produced for the whole stack by one strided
        # transposing copy each from the optimizer's flat bf16 shadow
        w2s = [weight_cache.get(params[i * PER_BLOCK + 9], adt) for i in range(nblk)]
        fuse_mlp_bwd = gemm.ENABLED and gemm.FUSE_GELU_BWD and adt == torch.bfloat16 and dh.is_cuda
        if fuse_mlp_bwd:
            # fc2^T (nblk, 4C, C), proj^T (nblk, C, C) and -- fc1 / qkv input gradients as TN products on the LDS-DMA ring kernel (N = 384
            # columns, K = 1536 / 1152: its regime) -- fc1^T (nblk, C, 4C), qkv^T (nblk, C, 3C): ONE transposing launch for the four
            W2T, WPT, W1T, WQT = gemm.stacked_transposes(
                [w2s] + [[weight_cache.get(params[i * PER_BLOCK + k], adt) for i in range(nblk)] for k in (3, 7, 2)])
        dma_bwd = fuse_mlp_bwd and gemm.dma_supported(dh.reshape(R, C), W2T[0])
        bm_bwd = gemm.dma_bm(R)
        PGL = torch.empty(nblk, ((R + bm_bwd - 1) // bm_bwd if dma_bwd else gemm.tile_rows(R)) if fuse_mlp_bwd
                          else lib.gm3d_gelu_partial_rows(R), 4 * C, dtype=torch.float32, device=dev)
        SGL = torch.empty(nblk, 4 * C, dtype=torch.float32, device=dev)
        dp2_last = meta["dp"][nblk - 1][1]
        G, d_o, _ = residual_ln_bwd(dh, None, xf, mf, rf, final_w, dp2_last, T, None, True, adt, R, dy=DO[nblk - 1],
                                    partial=PLN[2 * nblk])
        g_final_w, g_final_b, db2 = SLN[2 * nblk, 0], SLN[2 * nblk, 1], SLN[2 * nblk, 2]
        # gradient of the positional embedding = sum over blocks of the gradient at every u = x + pos site.  The first site's gradient
        # (G of the last block) starts the sum: that buffer's only other reader, the next block's LayerNorm-2 backward, has run
        # by the time the next site accumulates into it -- no zero fill, no read-modify-write for the first site.
        dpos = torch.zeros(R, C, dtype=torch.float32, device=dev) if nblk == 1 else None
        for i in range(nblk - 1, -1, -1):
            ln1w, ln1b, wqkv, wproj, bproj, ln2w, ln2b, w1, b1, w2, b2 = params[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            u, m1, r1, qkv, lse, x1, m2, r2, f = saved[i * 9:(i + 1) * 9]
            dp1 = meta["dp"][i][0]
            dp2_prev = meta["dp"][i - 1][1] if i > 0 else None
            gi = grads[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            # mlp branch: x2 = x1 + dp2 * (g @ W2^T + b2)
            gi[10] = db2
            if fuse_mlp_bwd:
                df = (gemm.linear_gelu_bwd_dma(d_o, W2T[i], f, b1, DF[i], PGL[i], bm=bm_bwd) if dma_bwd
                      else gemm.linear_gelu_bwd(d_o, W2T[i], f, b1, DF[i], PGL[i]))
            else:
                dg = d_o @ weight_cache.get(w2, adt)
                df, _ = bias_gelu_bwd(dg, f, b1, adt, df=DF[i], partial=PGL[i])
            gi[8] = SGL[i]
            dh2 = gemm.mm(df, W1T[i]) if fuse_mlp_bwd else df @ weight_cache.get(w1, adt)
            dx1, d_p, _ = residual_ln_bwd(dh2, G, x1, m2, r2, ln2w, dp1, T, None, True, adt, R, dy=DP[i],
                                          partial=PLN[2 * i + 1])
            gi[5], gi[6], gi[4] = SLN[2 * i + 1, 0], SLN[2 * i + 1, 1], SLN[2 * i + 1, 2]
            # attention branch: x1 = u + dp1 * (a @ Wproj^T + bproj)
            da = gemm.mm(d_p, WPT[i]) if fuse_mlp_bwd else d_p @ weight_cache.get(wproj, adt)
            dqkv = _attention_bwd(qkv, A[i], da, lse, B, T, H, scale, dqkv=DQ[i])
            dh1 = gemm.mm(dqkv, WQT[i]) if fuse_mlp_bwd else dqkv @ weight_cache.get(wqkv, adt)
            last = DIRECT_INPUT_GRADS and i == 0 and nblk > 1 and adt != torch.float32 and ctx.in_dtypes == (adt, adt)
            if last:
                # the stack's input gradients in the inputs' own (activation) type, written by this last pass instead of two
                # converting copies behind it: dy = 1 * dx (no DropPath factor in front of block 0), acc_out = the finished dpos
                dx_a = torch.empty(R, C, dtype=adt, device=dev)
                dpos_a = torch.empty(R, C, dtype=adt, device=dev)
                G, _, _ = residual_ln_bwd(dh1, dx1, u, m1, r1, ln1w, None, T, dpos, True, adt, R, dy=dx_a, partial=PLN[0],
                                          acc_out=dpos_a)
            else:
                G, d_o, _ = residual_ln_bwd(dh1, dx1, u, m1, r1, ln1w, dp2_prev, T, dpos, i > 0, adt, R,
                                            dy=DO[i - 1] if i > 0 else None, partial=PLN[2 * i])
            if dpos is None:
                dpos = G
            gi[0], gi[1] = SLN[2 * i, 0], SLN[2 * i, 1]
            db2 = SLN[2 * i, 2]
            grads[i * PER_BLOCK:(i + 1) * PER_BLOCK] = gi
        finish_batched(PLN, SLN.view(2 * nblk + 1, 3 * C))
        finish_batched(PGL, SGL)
        # all weight gradients of the stack: 4 batched GEMMs
        # ... written straight into the parameters' slots of the optimizer's flat gradient buffer when the same-kind weights of
        # the stack are adjacent there (optim._kind_key): autograd then adopts the slot views as .grad, nothing is copied later
        from .optim import grad_slots
        def slot(k):     # only when no gradient has been accumulated into these parameters yet (the slot is overwritten)
            ps = [params[i * PER_BLOCK + k] for i in range(nblk)]
            return grad_slots.stacked(ps) if dev.type == "cuda" and all(p.grad is None for p in ps) else None
        # Inside an `async_wgrad` region (the engine's backward) the weight-gradient GEMMs leave the main stream:
        #  * the 12-block encoder's four go to a side stream at once: what follows them in the backward is the mini-PointNet's
        #    streaming (HBM-bound) passes, which leave the matrix cores idle;
        #  * the shallow decoders' are DEFERRED to that same point (beside the encoder's latency-bound input-gradient chain, which
        #    comes right after them, they cost 3.7 %): possible because their destinations are slots of the flat gradient buffer,
        #    so the views handed to autograd now are filled later.  The region's exit runs whatever is still deferred and joins.
        reg = _ASYNC_WGRAD
        fresh = dev.type == "cuda" and all(params[i * PER_BLOCK + k].grad is None for i in range(nblk) for k in (9, 7, 3, 2))
        ws = reg["stream"] if fresh else None
        slots = [slot(k) for k in (9, 7, 3, 2)] if fresh else [None] * 4
        jobs = [(DO, GG), (DF, H2), (DP, A), (DQ, H1)]
        own = [(dy_, x_, sl) for (dy_, x_), sl in zip(jobs, slots)]
        if ws is not None and nblk < reg["min_blocks"] and all(sl is not None for sl in slots):
            reg["deferred"].append(((DO, GG, DF, H2, DP, A, DQ, H1), own))
            gw2, gw1, gwp, gwq = slots
        else:
            if ws is not None and nblk >= reg["min_blocks"]:
                _streams.fork(ws, who="fused.TransformerStackFn.backward: weight-gradient stream (async_wgrad region)")
                reg["used"] = True
                for t in (DO, GG, DF, H2, DP, A, DQ, H1):
                    t.record_stream(ws)
            else:
                ws = None
            with (torch.cuda.stream(ws) if ws is not None else contextlib.nullcontext()):
                # the deferred decoders' requests and this stack's own: ONE launch (+ one slab sum) where the kernel takes them all
                gw2, gw1, gwp, gwq = _run_deferred(reg, own) if ws is not None else _wgrad_many(own)
        for i in range(nblk):
            grads[i * PER_BLOCK + 9], grads[i * PER_BLOCK + 7] = gw2[i], gw1[i]
            grads[i * PER_BLOCK + 3], grads[i * PER_BLOCK + 2] = gwp[i], gwq[i]
        if last:
            return (dx_a.view(B, T, C), dpos_a.view(B, T, C), None, g_final_w, g_final_b) + tuple(grads)
        dx = G.view(B, T, C).to(ctx.in_dtypes[0])
        return (dx, dpos.view(B, T, C).to(ctx.in_dtypes[1]), None, g_final_w, g_final_b) + tuple(grads)


def run_stack(blocks, final_norm, x, pos, training):
    """blocks: iterable of Block modules; final_norm: nn.LayerNorm.  x, pos (B,T,384)."""
    from . import models_mae_learn_loss as M  # drop_path_scale lives there (the tests replay recorded draws through it)
    blocks = list(blocks)
    adt = torch.bfloat16 if (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16) \
        or x.dtype == torch.bfloat16 else torch.float32
    B = x.shape[0]
    probs = []
    for b in blocks:
        p = getattr(b.drop_path, "drop_prob", 0.0)
        probs += [p, p]
    scales = M.drop_path_scales(B, probs, training, x.device)
    dp = [(scales[2 * i], scales[2 * i + 1]) for i in range(len(blocks))]
    meta = {"num_heads": blocks[0].attn.num_heads, "scale": blocks[0].attn.scale, "eps": blocks[0].norm1.eps,
            "final_eps": final_norm.eps, "adt": adt, "dp": dp, "grad": torch.is_grad_enabled()}
    params = []
    for b in blocks:
        params += block_params(b)
    ns = NOGRAD_SPLIT
    if (ns > 1 and x.is_cuda and not training and B % ns == 0 and B // ns >= 16
            and not (torch.is_grad_enabled() and (x.requires_grad or params[0].requires_grad))):
        # Inference-only stack (the EMA teacher): nothing couples the clouds of a batch, and even an 8192-row chain of these kernels
        # fills the chip about half -- the batch is cut into `ns` parts that run as parallel chains on their own streams (parallel
        # branches of a captured graph), each kernel on 1/ns of the rows.
        main = torch.cuda.current_stream()
        h = B // ns
        x, pos = x.contiguous(), pos.contiguous()
        outs = [None] * ns
        streams = [main] + [_split_stream(x.device, j) for j in range(1, ns)]
        for side in streams[1:]:
            _streams.fork(side, main, who="fused.run_stack: parallel inference chain (NOGRAD_SPLIT)")
            x.record_stream(side)
            pos.record_stream(side)
        if LOCKSTEP:
            # one launch of every chain in turn: the capture (and the graph executor's enqueue order) interleaves the branches;
            # every chain's final LayerNorm writes its rows of ONE output buffer (no concatenation afterwards)
            T_, C_ = x.shape[1], x.shape[2]
            out = torch.empty(B, T_, C_, dtype=adt, device=x.device)
            for side in streams[1:]:
                out.record_stream(side)
            with torch.autocast("cuda", enabled=False), torch.no_grad():
                gens = [TransformerStackFn._forward_gen(_NoCtx(), x[j * h:(j + 1) * h], pos[j * h:(j + 1) * h],
                                                        dict(meta, hout=out[j * h:(j + 1) * h].view(h * T_, C_)), final_norm.weight,
                                                        final_norm.bias, *params) for j in range(ns)]
                live = ns
                while live:
                    for j in range(ns):
                        if gens[j] is None:
                            continue
                        with torch.cuda.stream(streams[j]):
                            try:
                                next(gens[j])
                            except StopIteration as done:
                                outs[j], gens[j] = done.value, None
                                live -= 1
        else:
            for j in range(1, ns):
                with torch.cuda.stream(streams[j]):
                    outs[j] = TransformerStackFn.apply(x[j * h:(j + 1) * h], pos[j * h:(j + 1) * h], meta, final_norm.weight,
                                                       final_norm.bias, *params)
            outs[0] = TransformerStackFn.apply(x[:h], pos[:h], meta, final_norm.weight, final_norm.bias, *params)
        for j in range(1, ns):
            _streams.join(streams[j], main)
            outs[j].record_stream(main)
        return out if LOCKSTEP else torch.cat(outs, dim=0)
    return TransformerStackFn.apply(x, pos, meta, final_norm.weight, final_norm.bias, *params)


class _NoCtx:
    """stand-in for the autograd context in the inference-only lock-step driver (nothing is saved)."""
    needs_input_grad = ()

    def save_for_backward(self, *a):
        pass


LOCKSTEP = True          # interleave the launches of the parallel inference chains (measured: same speed as chain after chain)
NOGRAD_SPLIT = 2         # parallel chains of an inference-only stack (1 = off; measured 2 > 1 > 3 > 4; tests flip it)
_split_streams = {}


def _split_stream(device, j=1):
    key = (device.type, device.index, j)
    if key not in _split_streams:
        _split_streams[key] = torch.cuda.Stream(device=device)
    return _split_streams[key]
