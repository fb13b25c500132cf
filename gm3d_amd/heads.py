"""The small layers around the transformer stacks as hand-written autograd nodes whose every row reduction is one of
our own kernels (deterministic two-stage column sums; no PyTorch multi-block reduce_kernel, which is what makes
the whole step safe to replay as a hipGraph on this stack -- see GraphedPretrainStep):

  PosEmbedFn      pos_embed = Linear(3,128) -> GELU -> Linear(128,384)        (P/models_mae_learn_loss.py:104-108)
  LinearBiasFn    increase_dim_just_network_without_feature = Conv1d(384,96)   (P/:169-176, used at :662)
  LossPredHeadFn  increase_dim_2 = Conv1d(384,1024) -> BN1d -> LeakyReLU(0.2) -> Conv1d(1024,384), then mean(-1)
                  (P/:152-158, :668, :677), last conv + mean folded into one 1024-vector
  ExpandRowsFn    mask_token.expand(B, N, -1)                                  (P/:653)
  rank_loss       forward_learning_loss(relative=True)                          (P/:795-805)
"""
import torch

from ._capi import lib
from .embed import _c32, _finish, colsum, splitk_wgrad


def _defer_wgrad(dy, x, w=None):
    from . import fused
    return fused.defer_wgrad(dy, x, w)
from .fused import weight_cache
from . import gemm
from .ops import _launch, _ptr, _stream, _DT


def _adt():
    return torch.bfloat16 if (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16) \
        else torch.float32


def _finish64(part, nrows, ncols):
    out = torch.empty(ncols, dtype=torch.float64, device=part.device)
    _launch("gm3d_colsum_finish_f64", {"rows": nrows, "cols": ncols}, lib.gm3d_colsum_finish_f64, _ptr(part), nrows, ncols,
            ncols, _ptr(out), _stream())
    return out


def _ragged_ok(x2, W):
    """bf16 rows against a weight whose width is a multiple of 8 but not of 128 (the 96-wide reconstruction head): the ring
    kernel's ragged last column tile."""
    return (gemm.ENABLED and x2.is_cuda and x2.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and W.shape[0] % 8 == 0
            and W.shape[1] % 64 == 0 and W.stride(1) == 1 and W.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0)


class LinearBiasFn(torch.autograd.Function):
    """y = x @ W^T + b on rows; x (..,K), W (N,K[,1]) fp32 master, b (N) or None; computed in adt.  bf16: our own MFMA kernels
    forward and backward (gemm.mm / the ring kernel's ragged tile for widths like 96; input gradient as a TN product against
    W^T; a width below 128 is zero-padded ONCE for both backward products).  The bias gradient is our own column-sum kernel:
    PyTorch's multi-block reduction is not hipGraph-replay-safe on this stack."""

    @staticmethod
    def forward(ctx, x, w, b, adt):
        with torch.autocast("cuda", enabled=False):
            shp = x.shape
            x2 = x.reshape(-1, shp[-1]).to(adt).contiguous()
            W = weight_cache.get(w, adt).reshape(w.shape[0], -1)
            if gemm.supported(x2, W) or gemm.ragged_supported(x2, W):
                # gemm.mm names the kernel: the tiled families, the ring kernel's ragged last column tile (N = 96), csrc/gemm.hip's
                # ragged-K form (K = 96), the weight-stationary kernel for the tall ragged shapes
                y = gemm.mm(x2, W, _c32(b) if b is not None else None)
            elif _ragged_ok(x2, W):
                y = gemm.linear_tn_ring(x2, W, _c32(b) if b is not None else None)
            else:
                y = torch.addmm(weight_cache.get(b, adt), x2, W.t()) if b is not None else x2 @ W.t()
            ctx.save_for_backward(x2, w)
            ctx.adt, ctx.shp, ctx.xdt, ctx.has_bias = adt, shp, x.dtype, b is not None
            return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        with torch.autocast("cuda", enabled=False):
            x2, w = ctx.saved_tensors
            adt = ctx.adt
            dy2 = dy.reshape(-1, dy.shape[-1]).to(adt).contiguous()
            W = weight_cache.get(w, adt).reshape(w.shape[0], -1)
            N, K = W.shape
            R = dy2.shape[0]
            db = colsum(dy2, adt) if ctx.has_bias else None
            if (adt == torch.bfloat16 and N < 128 and K % 128 == 0 and not gemm.RAGGED
                    and gemm.wgrad_supported(dy2.new_empty(1, R, 128), x2.unsqueeze(0))):
                # narrow layer (N = 96) without the ragged-width kernels: dy and W^T zero-padded to one 128-wide tile, shared by both products
                pad = torch.empty(R, 128, dtype=adt, device=dy2.device)
                _launch("gm3d_pad_cols", {"R": R, "N": N}, lib.gm3d_pad_cols, _ptr(dy2), dy2.stride(0), R, N, _ptr(pad), 128, _DT[adt], _stream())
                wt = torch.zeros(K, 128, dtype=adt, device=dy2.device)
                wt[:, :N] = W.t()
                dx = gemm.mm(pad, wt).view(ctx.shp).to(ctx.xdt)
                dW = gemm.wgrad_nt(pad.unsqueeze(0), x2.unsqueeze(0))[0][:N].contiguous().reshape(w.shape)
                return dx, dW, db, None
            dx = gemm.mm_nn(dy2, W).view(ctx.shp).to(ctx.xdt)
            dW = _defer_wgrad(dy2, x2, w).reshape(w.shape)
            return dx, dW, db, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (4 <= C <= 512, C % 4 == 0) on csrc/rowops.hip's plain LayerNorm: x in any float type
    is normalised in `adt` storage with fp32 statistics -- one launch forward, one + one finish backward, no autocast casts, and the
    gamma / beta gradients by our own deterministic column sums (hipGraph-replay-safe)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, adt):
        with torch.autocast("cuda", enabled=False):
            C = x.shape[-1]
            x2 = x.reshape(-1, C).to(adt).contiguous()
            R = x2.shape[0]
            h = torch.empty_like(x2)
            mean = torch.empty(R, dtype=torch.float32, device=x.device)
            rstd = torch.empty(R, dtype=torch.float32, device=x.device)
            wc, bc = _c32(w), _c32(b)
            _launch("gm3d_ln_plain_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_ln_plain_fwd, _ptr(x2), _ptr(wc), _ptr(bc),
                    float(eps), _ptr(h), _ptr(mean), _ptr(rstd), R, C, _DT[adt], _stream())
            ctx.save_for_backward(x2, mean, rstd, w)
            ctx.adt, ctx.shp, ctx.xdt = adt, x.shape, x.dtype
            return h.view(x.shape)

    @staticmethod
    def backward(ctx, dh):
        with torch.autocast("cuda", enabled=False):
            x2, mean, rstd, w = ctx.saved_tensors
            adt = ctx.adt
            R, C = x2.shape
            dh2 = dh.reshape(R, C).to(adt).contiguous()
            dx = torch.empty_like(x2)
            nrows = lib.gm3d_ln_plain_partial_rows(R)
            part = torch.empty(nrows, 2 * C, dtype=torch.float32, device=dh.device)
            _launch("gm3d_ln_plain_bwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_ln_plain_bwd, _ptr(dh2), _ptr(x2), _ptr(mean),
                    _ptr(rstd), _ptr(_c32(w)), _ptr(dx), _ptr(part), R, C, _DT[adt], _stream())
            gb = _finish(part, nrows, 2 * C)
            return dx.view(ctx.shp).to(ctx.xdt), gb[:C].to(w.dtype), gb[C:].to(w.dtype), None, None


class AddLayerNormFn(torch.autograd.Function):
    """(s, h) with s = x + rowscale[sample] * (y + ybias) + z and h = LayerNorm(s): the residual sum(s) in front of a pre-norm block's
    LayerNorm formed inside the LayerNorm kernel (csrc/rowops.hip gm3d_add_ln_fwd/bwd; any width <= 512).  y, ybias, rowscale, z
    optional.  Backward: ONE launch gives the gradient of s (= of x and z), of y (scaled), LayerNorm's gamma / beta sums and the
    column sum that is ybias's gradient (the bias of the Linear that produced y is added here, so that Linear runs without one)."""

    @staticmethod
    def forward(ctx, x, y, ybias, rowscale, z, w, b, eps, adt):
        with torch.autocast("cuda", enabled=False):
            C = x.shape[-1]
            x2 = x.reshape(-1, C).to(adt).contiguous()
            R = x2.shape[0]
            y2 = y.reshape(R, C).to(adt).contiguous() if y is not None else None
            z2 = z.reshape(R, C).to(adt).contiguous() if z is not None else None
            rps = R // rowscale.shape[0] if rowscale is not None else 1
            h = torch.empty_like(x2)
            s = torch.empty_like(x2) if (y2 is not None or z2 is not None) else None
            mean = torch.empty(R, dtype=torch.float32, device=x.device)
            rstd = torch.empty(R, dtype=torch.float32, device=x.device)
            yb = _c32(ybias) if ybias is not None else None
            _launch("gm3d_add_ln_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_fwd, _ptr(x2), _ptr(y2), _ptr(yb), _ptr(rowscale),
                    int(rps), _ptr(z2), _ptr(_c32(w)), _ptr(_c32(b)), float(eps), _ptr(s), _ptr(h), _ptr(mean), _ptr(rstd), R, C, _DT[adt],
                    _stream())
            if s is None:        # plain LayerNorm: the "sum" is x itself (a copy: an autograd node must not hand an input back as an output)
                s = x2.clone()
            ctx.save_for_backward(s, mean, rstd, w, rowscale)
            ctx.adt, ctx.shp, ctx.rps = adt, x.shape, rps
            ctx.have = (y is not None, ybias is not None, z is not None)
            ctx.dts = (x.dtype, y.dtype if y is not None else None, z.dtype if z is not None else None, w.dtype)
            return s.view(x.shape), h.view(x.shape)

    @staticmethod
    def backward(ctx, ds, dh):
        with torch.autocast("cuda", enabled=False):
            s, mean, rstd, w, rowscale = ctx.saved_tensors
            adt = ctx.adt
            R, C = s.shape
            has_y, has_yb, has_z = ctx.have
            dh2 = dh.reshape(R, C).to(adt).contiguous() if dh is not None else torch.zeros(R, C, dtype=adt, device=s.device)
            gin = ds.reshape(R, C).to(adt).contiguous() if ds is not None else None
            dx = torch.empty_like(s)
            nsum = 3 if has_yb else 2
            dy = torch.empty_like(s) if (has_y and rowscale is not None) else None
            nrows = lib.gm3d_ln_plain_partial_rows(R)
            part = torch.empty(nrows, nsum * C, dtype=torch.float32, device=s.device)
            _launch("gm3d_add_ln_bwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_bwd, _ptr(dh2), _ptr(gin), _ptr(s), _ptr(mean),
                    _ptr(rstd), _ptr(_c32(w)), _ptr(rowscale) if has_y else None, int(ctx.rps), _ptr(dx), _ptr(dy), _ptr(part), nsum, R, C,
                    _DT[adt], _stream())
            gb = _finish(part, nrows, nsum * C)
            xdt, ydt, zdt, wdt = ctx.dts
            gx = dx.view(ctx.shp)
            gy = (dy if dy is not None else dx).view(ctx.shp).to(ydt) if has_y else None
            return (gx.to(xdt), gy, gb[2 * C:].clone() if has_yb else None, None, gx.to(zdt) if has_z else None, gb[:C].to(wdt),
                    gb[C:2 * C].to(wdt), None, None)


class TakeRowsFn(torch.autograd.Function):
    """y[b, j] = x[b, ids[b, j]] for ids (B,J) int64 that may REPEAT a source row (a token that is a member of several groups; the
    three nearest coarse tokens of the token propagation).  Forward: torch.gather.  Backward: csrc/gather.hip -- the inverse lists of
    ids in ascending j, then every source row sums its readers' gradients in that order: deterministic, unlike the colliding float
    atomics of PyTorch's gather backward."""

    @staticmethod
    def forward(ctx, x, ids):
        B, S, C = x.shape
        ids = ids.contiguous()
        ctx.save_for_backward(ids)
        ctx.S, ctx.xdt = S, x.dtype
        if x.element_size() in (2, 4):
            x = x.contiguous()
            out = torch.empty(B, ids.shape[1], C, dtype=x.dtype, device=x.device)
            _launch("gm3d_take_rows", {"B": B, "J": ids.shape[1], "C": C}, lib.gm3d_take_rows, _ptr(x), _ptr(ids), _ptr(out), B, S,
                    ids.shape[1], C, x.element_size(), _stream())
            return out
        return torch.gather(x, 1, ids.unsqueeze(-1).expand(-1, -1, C))

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        B, J = ids.shape
        S, C = ctx.S, dy.shape[-1]
        dy = dy.contiguous()
        if dy.dtype not in _DT:
            dy = dy.float()
        off = torch.empty(B, S + 1, dtype=torch.int32, device=dy.device)
        lst = torch.empty(B, J, dtype=torch.int32, device=dy.device)
        _launch("gm3d_gather_inverse", {"B": B, "J": J, "S": S}, lib.gm3d_gather_inverse, _ptr(ids), B, J, S, _ptr(off), _ptr(lst), _stream())
        dx = torch.empty(B, S, C, dtype=dy.dtype, device=dy.device)
        _launch("gm3d_gather_rows_bwd", {"B": B, "J": J, "S": S, "C": C, "dtype": str(dy.dtype)}, lib.gm3d_gather_rows_bwd, _ptr(dy), _ptr(off),
                _ptr(lst), _ptr(dx), B, J, S, C, _DT[dy.dtype], _stream())
        return dx.to(ctx.xdt), None


class GroupMaxFn(torch.autograd.Function):
    """x (G,K,C) -> (G,C): max over the K members of a group (the pooling of a mini-PointNet token embed, P/:895,898
    `torch.max(.., dim)[0]`) on csrc/embed.hip's group_max kernels: the winner is the FIRST maximising member (uint8 index), the
    backward hands the gradient to that member only -- torch.max's convention, the one the level-0 embed (embed.EmbedFn) uses."""

    @staticmethod
    def forward(ctx, x):
        G, K, C = x.shape
        x = x.contiguous()
        out = torch.empty(G, C, dtype=x.dtype, device=x.device)
        arg = torch.empty(G, C, dtype=torch.uint8, device=x.device)
        _launch("gm3d_group_max_fwd", {"G": G, "K": K, "C": C, "dtype": str(x.dtype)}, lib.gm3d_group_max_fwd, _ptr(x), None, _ptr(out),
                _ptr(arg), G, K, C, _DT[x.dtype], _stream())
        ctx.save_for_backward(arg)
        ctx.dims = (G, K, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        G, K, C = ctx.dims
        dout = dout.contiguous()
        dx = torch.empty(G, K, C, dtype=dout.dtype, device=dout.device)
        _launch("gm3d_group_max_bwd", {"G": G, "K": K, "C": C, "dtype": str(dout.dtype)}, lib.gm3d_group_max_bwd, _ptr(dout), _ptr(arg),
                _ptr(dx), G, K, C, _DT[dout.dtype], _stream())
        return dx


def group_max_supported(x):
    return x.is_cuda and x.dim() == 3 and x.dtype in _DT and x.shape[2] % 8 == 0 and 1 <= x.shape[1] <= 255 and x.shape[2] <= 1024


class Interp3Fn(torch.autograd.Function):
    """[fine | sum_j w_j * coarse[idx_j]] -> (B,N,C1+C2): the token propagation's 3-NN interpolation and concatenation as one launch
    (csrc/gather.hip gm3d_interp3_fwd); backward: the fine half is a slice, the coarse tokens sum their readers' weighted gradients in
    ascending reader order (gm3d_gather_inverse + gm3d_gather_rows_bwd_w: deterministic)."""

    @staticmethod
    def forward(ctx, fine, coarse, idx, w):
        B, N, C1 = fine.shape
        S, C2 = coarse.shape[1], coarse.shape[2]
        fine, coarse = fine.contiguous(), coarse.to(fine.dtype).contiguous()
        idx, w = idx.contiguous(), w.float().contiguous()
        out = torch.empty(B, N, C1 + C2, dtype=fine.dtype, device=fine.device)
        _launch("gm3d_interp3_fwd", {"B": B, "N": N, "C": C1 + C2, "dtype": str(fine.dtype)}, lib.gm3d_interp3_fwd, _ptr(coarse), _ptr(idx),
                _ptr(w), _ptr(fine), _ptr(out), B, N, S, C1, C2, _DT[fine.dtype], _stream())
        ctx.save_for_backward(idx, w)
        ctx.dims = (B, N, S, C1, C2)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        B, N, S, C1, C2 = ctx.dims
        dout = dout.contiguous()
        J = 3 * N
        off = torch.empty(B, S + 1, dtype=torch.int32, device=dout.device)
        lst = torch.empty(B, J, dtype=torch.int32, device=dout.device)
        _launch("gm3d_gather_inverse", {"B": B, "J": J, "S": S}, lib.gm3d_gather_inverse, _ptr(idx), B, J, S, _ptr(off), _ptr(lst), _stream())
        dcoarse = torch.empty(B, S, C2, dtype=dout.dtype, device=dout.device)
        _launch("gm3d_gather_rows_bwd", {"B": B, "J": J, "S": S, "C": C2, "dtype": str(dout.dtype)}, lib.gm3d_gather_rows_bwd_w, _ptr(dout),
                C1 + C2, C1, 3, _ptr(w), _ptr(off), _ptr(lst), _ptr(dcoarse), B, J, S, C2, _DT[dout.dtype], _stream())
        return dout[..., :C1], dcoarse, None, None


class WhereRowsFn(torch.autograd.Function):
    """out[b][t] = masked[b][t] ? alt[b][t] : a[b][t]  (a (B,T,C); alt (B,T,C), a single row (1,1,C) / (C,) -- the mask token --, or None
    = zeros; masked (B,T) bool): the hierarchical model's per-token choices as one launch forward and one per input backward
    (csrc/gather.hip gm3d_where_rows).  A broadcast alt's gradient is a column sum over the masked rows (our two-stage kernel)."""

    @staticmethod
    def forward(ctx, masked, a, alt):
        B, T, C = a.shape
        a = a.contiguous()
        m = masked.contiguous()
        m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
        bcast = alt is not None and alt.numel() == C
        altc = None
        if alt is not None:
            altc = (weight_cache.get(alt, a.dtype) if bcast else alt.to(a.dtype)).contiguous()
        out = torch.empty_like(a)
        _launch("gm3d_where_rows", {"rows": B * T, "C": C}, lib.gm3d_where_rows, _ptr(m), 0, _ptr(a), _ptr(altc), int(bcast), _ptr(out), B * T, C,
                a.element_size(), _stream())
        ctx.save_for_backward(m)
        ctx.info = (B, T, C, bcast, alt.dtype if alt is not None else None, alt.shape if alt is not None else None, a.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        (m,) = ctx.saved_tensors
        B, T, C, bcast, altdt, altshape, adt = ctx.info
        dout = dout.contiguous()
        da = dalt = None
        if ctx.needs_input_grad[1]:
            da = torch.empty_like(dout)
            _launch("gm3d_where_rows", {"rows": B * T, "C": C}, lib.gm3d_where_rows, _ptr(m), 0, _ptr(dout), None, 0, _ptr(da), B * T, C,
                    dout.element_size(), _stream())
        if altshape is not None and ctx.needs_input_grad[2]:
            dm = torch.empty_like(dout)
            _launch("gm3d_where_rows", {"rows": B * T, "C": C}, lib.gm3d_where_rows, _ptr(m), 1, _ptr(dout), None, 0, _ptr(dm), B * T, C,
                    dout.element_size(), _stream())
            if bcast:      # (our two-stage column sum takes widths that are multiples of 8: every token width of the model)
                dalt = (colsum(dm.view(B * T, C), dm.dtype) if C % 8 == 0 else dm.float().sum(dim=(0, 1))).view(altshape).to(altdt)
            else:
                dalt = dm.to(altdt)
        return None, da, dalt


def where_rows(masked, a, alt=None):
    """masked ? alt : a per token, one launch (GPU, 2- or 4-byte elements); torch.where elsewhere."""
    if a.is_cuda and a.dim() == 3 and a.element_size() in (2, 4) and masked.shape == a.shape[:2]:
        return WhereRowsFn.apply(masked, a, alt)
    m = masked.unsqueeze(-1)
    if alt is None:
        return torch.where(m, torch.zeros((), dtype=a.dtype, device=a.device), a)
    return torch.where(m, alt.to(a.dtype).reshape(1, 1, -1) if alt.numel() == a.shape[-1] else alt.to(a.dtype), a)


def take_rows(x, ids):
    """x (B,S,C) gathered along dim 1 by ids (B,J) with a deterministic backward (GPU, C % 8 == 0); else models_mae_learn_loss.take."""
    if x.is_cuda and x.dim() == 3 and x.shape[-1] % 8 == 0 and ids.dtype == torch.int64 and x.shape[1] <= 4096 and ids.shape[1] <= 16384:
        return TakeRowsFn.apply(x, ids)
    from .models_mae_learn_loss import take
    return take(x, ids)


class ResidualTailFn(torch.autograd.Function):
    """out = x + rowscale[sample] * (y + ybias): the last residual sum of a block stack (no LayerNorm behind it) on the same
    kernels as AddLayerNormFn -- the gradient of ybias is our own column sum (PyTorch's reduction over B*T rows is not
    hipGraph-replay-safe on this stack: DESIGN 3c; it came back non-finite from the second replay of the Point-M2AE step)."""

    @staticmethod
    def forward(ctx, x, y, ybias, rowscale, adt):
        with torch.autocast("cuda", enabled=False):
            C = x.shape[-1]
            x2 = x.reshape(-1, C).to(adt).contiguous()
            R = x2.shape[0]
            y2 = y.reshape(R, C).to(adt).contiguous()
            rps = R // rowscale.shape[0] if rowscale is not None else 1
            s = torch.empty_like(x2)
            _launch("gm3d_add_ln_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_fwd, _ptr(x2), _ptr(y2), _ptr(_c32(ybias)),
                    _ptr(rowscale), int(rps), None, None, None, 0.0, _ptr(s), None, None, None, R, C, _DT[adt], _stream())
            ctx.save_for_backward(rowscale)
            ctx.adt, ctx.shp, ctx.rps, ctx.dts = adt, x.shape, rps, (x.dtype, y.dtype)
            return s.view(x.shape)

    @staticmethod
    def backward(ctx, ds):
        with torch.autocast("cuda", enabled=False):
            (rowscale,) = ctx.saved_tensors
            adt = ctx.adt
            C = ctx.shp[-1]
            gin = ds.reshape(-1, C).to(adt).contiguous()
            R = gin.shape[0]
            dx = torch.empty_like(gin)
            dy = torch.empty_like(gin) if rowscale is not None else None
            nrows = lib.gm3d_ln_plain_partial_rows(R)
            part = torch.empty(nrows, 3 * C, dtype=torch.float32, device=gin.device)
            # dh == NULL: the sum-only backward reads nothing of a LayerNorm that is not there (dx = gin, an Inf stays an Inf)
            _launch("gm3d_add_ln_bwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_bwd_acc, None, _ptr(gin), None, None,
                    None, None, _ptr(rowscale), int(ctx.rps), _ptr(dx), _ptr(dy), _ptr(part), 3, None, 0, R, C, _DT[adt], _stream())
            gb = _finish(part, nrows, 3 * C)
            xdt, ydt = ctx.dts
            return dx.view(ctx.shp).to(xdt), (dy if dy is not None else dx).view(ctx.shp).to(ydt), gb[2 * C:].clone(), None, None


class BiasGeluFn(torch.autograd.Function):
    """g = GELU(f + bias) (exact erf) on csrc/rowops.hip's streaming kernels, any width % 8 == 0; the backward also yields the bias
    gradient (fc1's: that Linear runs without a bias)."""

    @staticmethod
    def forward(ctx, f, bias, adt):
        from .fused import bias_gelu_fwd
        with torch.autocast("cuda", enabled=False):
            C = f.shape[-1]
            f2 = f.reshape(-1, C).to(adt).contiguous()
            bf = _c32(bias)
            g = bias_gelu_fwd(f2, bf, adt)
            ctx.save_for_backward(f2, bias)
            ctx.adt, ctx.shp, ctx.fdt = adt, f.shape, f.dtype
            return g.view(f.shape)

    @staticmethod
    def backward(ctx, dg):
        from .fused import bias_gelu_bwd
        with torch.autocast("cuda", enabled=False):
            f2, bias = ctx.saved_tensors
            adt = ctx.adt
            dg2 = dg.reshape(f2.shape).to(adt).contiguous()
            df, db = bias_gelu_bwd(dg2, f2, _c32(bias), adt)
            return df.view(ctx.shp).to(ctx.fdt), db.to(bias.dtype), None


def layer_norm_supported(x, C):
    return x.is_cuda and 4 <= C <= 512 and C % 4 == 0 and x.dtype in (torch.float32, torch.bfloat16)


class PosEmbedFn(torch.autograd.Function):
    """center (B,G,3) f32 -> (B,G,384) adt.  The K=3 layer and its GELU are one streaming kernel; its backward is a
    pure reduction (xyz carries no gradient)."""

    @staticmethod
    def forward(ctx, center, w0, b0, w1, b1, adt):
        with torch.autocast("cuda", enabled=False):
            B, G, _ = center.shape
            R, C = B * G, w0.shape[0]
            x = center.reshape(R, 3).float().contiguous()
            w0f, b0f = _c32(w0), _c32(b0)
            h = torch.empty(R, C, dtype=adt, device=x.device)
            _launch("gm3d_lin3_gelu_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_lin3_gelu_fwd, _ptr(x), _ptr(w0f),
                    _ptr(b0f), _ptr(h), R, C, _DT[adt], _stream())
            W1 = weight_cache.get(w1, adt)
            out = gemm.mm(h, W1, _c32(b1))
            ctx.save_for_backward(x, h, w0f, b0f, w1)
            ctx.adt, ctx.dims = adt, (B, G, R, C)
            return out.view(B, G, -1)

    @staticmethod
    def backward(ctx, dout):
        with torch.autocast("cuda", enabled=False):
            x, h, w0f, b0f, w1 = ctx.saved_tensors
            adt = ctx.adt
            B, G, R, C = ctx.dims
            d = dout.reshape(R, -1).to(adt).contiguous()
            W1 = weight_cache.get(w1, adt)
            dW1 = _defer_wgrad(d, h, w1)
            db1 = colsum(d, adt)
            dh = gemm.mm_nn(d, W1).contiguous()
            nrows = lib.gm3d_embed_partial_rows(3, R, C)
            part = torch.empty(nrows, 4 * C, dtype=torch.float64, device=x.device)
            _launch("gm3d_lin3_gelu_bwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_lin3_gelu_bwd, _ptr(dh), _ptr(x),
                    _ptr(w0f), _ptr(b0f), R, C, _ptr(part), _DT[adt], _stream())
            dW0 = torch.empty(C, 3, dtype=torch.float32, device=x.device)
            db0 = torch.empty(C, dtype=torch.float32, device=x.device)
            _launch("gm3d_colsum_finish_f64", {"rows": nrows, "cols": 4 * C}, lib.gm3d_lin3_finish, _ptr(part), nrows, C, _ptr(dW0), _ptr(db0),
                    _stream())
            return None, dW0, db0, dW1, db1, None


class LossPredHeadFn(torch.autograd.Function):
    """x (B,L,384) -> (B,L) f32: Conv1d(384,1024) -> BatchNorm1d (batch statistics over B*L rows in train mode) ->
    LeakyReLU(0.2) -> [Conv1d(1024,384) ; mean over its 384 outputs] as one 1024-vector."""

    @staticmethod
    def forward(ctx, x, w0, b0, gamma, beta, w1, b1, rm, rv, nbt, meta):
        with torch.autocast("cuda", enabled=False):
            adt, training, eps, mom, slope = meta["adt"], meta["training"], meta["eps"], meta["momentum"], meta["slope"]
            B, L, Cin = x.shape
            R, C = B * L, w0.shape[0]
            K = 32 if R % 32 == 0 else 1
            G = R // K
            dev = x.device
            x2 = x.reshape(R, Cin).to(adt).contiguous()
            W0 = weight_cache.get(w0, adt).reshape(C, Cin)
            y0 = gemm.mm(x2, W0)
            t = weight_cache.get(b0, adt).detach().unsqueeze(0).expand(G, C).contiguous()
            st = None
            if training:
                nrows = lib.gm3d_embed_partial_rows(1, G, C)
                part = torch.empty(nrows, 2 * C, dtype=torch.float32, device=dev)
                _launch("gm3d_bn_bcast_stats", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_stats, _ptr(y0),
                        _ptr(t), G, K, C, _ptr(part), _DT[adt], _stream())
                st = _finish(part, nrows, 2 * C)
            f32 = dict(dtype=torch.float32, device=dev)
            gc, bc = _c32(gamma), _c32(beta)
            scale, shift, mean, rstd = (torch.empty(C, **f32) for _ in range(4))
            _launch("gm3d_bn_finalize", {"C": C}, lib.gm3d_bn_finalize, _ptr(st), float(R), _ptr(gc), _ptr(bc), float(eps),
                    float(mom), _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), C,
                    int(training), _stream())
            a = torch.empty(R, C, dtype=adt, device=dev)
            _launch("gm3d_bn_bcast_apply_relu", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_apply_relu,
                    _ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(a), G, K, C, float(slope), _DT[adt], _stream())
            if meta.get("act_taps") is not None:          # a test records the LeakyReLU's sign pattern (leaky keeps the sign)
                meta["act_taps"].append(a.detach() > 0)
            W1 = _c32(w1.reshape(w1.shape[0], C))
            nout = w1.shape[0]
            wv, wv_t, bm = torch.empty(C, **f32), torch.empty(C, dtype=adt, device=dev), torch.empty(1, **f32)
            _launch("gm3d_head_fold", {"C": C}, lib.gm3d_head_fold, _ptr(W1), _ptr(_c32(b1)), nout, C, _ptr(wv), _ptr(wv_t), _ptr(bm),
                    _DT[adt], _stream())                            # wv = mean over the 384 output rows of W1, bm = mean(b1)
            out = torch.empty(R, **f32)
            _launch("gm3d_head_rowdot", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_head_rowdot, _ptr(a), _ptr(wv_t), _ptr(bm), R, C,
                    _ptr(out), _DT[adt], _stream())
            if meta["grad"] and any(ctx.needs_input_grad):
                if not training:
                    raise NotImplementedError("LossPredHeadFn backward is implemented for train-mode BatchNorm only")
                ctx.save_for_backward(x2, y0, t, a, w0, gamma, w1, wv, mean, rstd, scale, shift)
                ctx.meta, ctx.dims, ctx.xdt = meta, (B, L, Cin, R, C, G, K), x.dtype
            return out.view(B, L)

    @staticmethod
    def backward(ctx, dout):
        with torch.autocast("cuda", enabled=False):
            x2, y0, t, a, w0, gamma, w1, wv, mean, rstd, scale, shift = ctx.saved_tensors
            meta = ctx.meta
            adt, slope = meta["adt"], meta["slope"]
            B, L, Cin, R, C, G, K = ctx.dims
            dev = dout.device
            d = dout.reshape(R).float().contiguous()
            nout = w1.shape[0]
            # out = a @ wv + mean(b1), wv = mean_rows(W1)
            # dwv (C,) = a^T d: a weighted column sum in two deterministic stages (was a library GEMV)
            nrows = lib.gm3d_embed_partial_rows(2, R, C)
            part = torch.empty(nrows, C, dtype=torch.float32, device=dev)
            _launch("gm3d_colsum_partial_w", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_colsum_partial_w, _ptr(a), _ptr(d), R, C,
                    _ptr(part), _DT[adt], _stream())
            dwv = _finish(part, nrows, C)
            dW1 = torch.empty(w1.shape, dtype=torch.float32, device=dev)
            db1 = torch.empty(nout, dtype=torch.float32, device=dev)
            _launch("gm3d_head_fold_bwd", {"C": C}, lib.gm3d_head_fold_bwd, _ptr(dwv), _ptr(d), R, nout, C, _ptr(dW1), _ptr(db1),
                    _stream())
            da = torch.empty(R, C, dtype=adt, device=dev)                            # (R,C) = d (x) wv
            _launch("gm3d_head_outer", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_head_outer, _ptr(d), _ptr(wv), R, C, _ptr(da),
                    _DT[adt], _stream())
            nrows = lib.gm3d_embed_partial_rows(1, G, C)
            part = torch.empty(nrows, 2 * C, dtype=torch.float32, device=dev)
            _launch("gm3d_bn_bcast_bwd_stats", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_bwd_stats,
                    _ptr(da), _ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), G, K, C, _ptr(part),
                    float(slope), _DT[adt], _stream())
            s12 = _finish(part, nrows, 2 * C)
            s1, s2 = s12[:C], s12[C:]
            dy = torch.empty(R, C, dtype=adt, device=dev)
            dt = torch.empty(G, C, dtype=torch.float32, device=dev)
            _launch("gm3d_bn_bcast_bwd_apply", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_bwd_apply,
                    _ptr(da), _ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(s1), _ptr(s2),
                    _ptr(dy), _ptr(dt), G, K, C, float(slope), _DT[adt], _stream())
            W0 = weight_cache.get(w0, adt).reshape(C, Cin)
            dx = gemm.mm_nn(dy, W0).view(B, L, Cin).to(ctx.xdt)
            dW0 = _defer_wgrad(dy, x2, w0).reshape(w0.shape)
            db0 = colsum(dt, torch.float32)
            return dx, dW0, db0, s2, s1, dW1, db1, None, None, None, None


class BnBcastActFn(torch.autograd.Function):
    """a = act(BatchNorm1d(y0 + t[group])) on a (G*K, C) row layout: y0 (G*K, C) and the per-group term t (G, C) in `adt` (a
    Conv1d bias in front of the BatchNorm is the constant case of t), act = ReLU / LeakyReLU(slope).  The streaming passes of the
    mini-PointNet embed (csrc/embed.hip bn_bcast_*) as an autograd node of their own: batch statistics, running-statistic update,
    normalise + activate forward; the two-pass BatchNorm backward with the gamma / beta sums by our own kernels.
    meta: adt, training, eps, momentum, slope, K, grad."""

    @staticmethod
    def forward(ctx, y0, t, gamma, beta, rm, rv, nbt, meta):
        with torch.autocast("cuda", enabled=False):
            adt, training, K, slope = meta["adt"], meta["training"], meta["K"], meta["slope"]
            G, C = t.shape
            R = G * K
            dev = y0.device
            ctx.in_dtypes = (y0.dtype, t.dtype)
            y0 = y0.reshape(R, C).to(adt).contiguous()
            t = t.to(adt).contiguous()
            st = None
            if training:
                nrows = lib.gm3d_embed_partial_rows(1, G, C)
                part = torch.empty(nrows, 2 * C, dtype=torch.float32, device=dev)
                _launch("gm3d_bn_bcast_stats", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_stats, _ptr(y0), _ptr(t),
                        G, K, C, _ptr(part), _DT[adt], _stream())
                st = _finish(part, nrows, 2 * C)
            f32 = dict(dtype=torch.float32, device=dev)
            scale, shift, mean, rstd = (torch.empty(C, **f32) for _ in range(4))
            _launch("gm3d_bn_finalize", {"C": C}, lib.gm3d_bn_finalize, _ptr(st), float(R), _ptr(_c32(gamma)), _ptr(_c32(beta)),
                    float(meta["eps"]), float(meta["momentum"]), _ptr(rm), _ptr(rv), _ptr(nbt), _ptr(scale), _ptr(shift), _ptr(mean),
                    _ptr(rstd), C, int(training), _stream())
            a = torch.empty(R, C, dtype=adt, device=dev)
            _launch("gm3d_bn_bcast_apply_relu", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_apply_relu, _ptr(y0),
                    _ptr(t), _ptr(scale), _ptr(shift), _ptr(a), G, K, C, float(slope), _DT[adt], _stream())
            if meta["grad"] and any(ctx.needs_input_grad):
                if not training:
                    raise NotImplementedError("BnBcastActFn backward is implemented for train-mode BatchNorm only")
                ctx.save_for_backward(y0, t, mean, rstd, scale, shift)
                ctx.meta, ctx.dims = meta, (G, K, C)
            return a

    @staticmethod
    def backward(ctx, da):
        with torch.autocast("cuda", enabled=False):
            y0, t, mean, rstd, scale, shift = ctx.saved_tensors
            meta = ctx.meta
            adt, slope = meta["adt"], meta["slope"]
            G, K, C = ctx.dims
            dev = da.device
            da = da.reshape(G * K, C).to(adt).contiguous()
            nrows = lib.gm3d_embed_partial_rows(1, G, C)
            part = torch.empty(nrows, 2 * C, dtype=torch.float32, device=dev)
            _launch("gm3d_bn_bcast_bwd_stats", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_bwd_stats, _ptr(da),
                    _ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), G, K, C, _ptr(part), float(slope), _DT[adt],
                    _stream())
            s12 = _finish(part, nrows, 2 * C)
            s1, s2 = s12[:C], s12[C:]
            dy = torch.empty(G * K, C, dtype=adt, device=dev)
            dt = torch.empty(G, C, dtype=torch.float32, device=dev)
            _launch("gm3d_bn_bcast_bwd_apply", {"G": G, "K": K, "C": C, "dtype": str(adt)}, lib.gm3d_bn_bcast_bwd_apply, _ptr(da),
                    _ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(s1), _ptr(s2), _ptr(dy), _ptr(dt), G,
                    K, C, float(slope), _DT[adt], _stream())
            return dy.to(ctx.in_dtypes[0]), dt.to(ctx.in_dtypes[1]), s2, s1, None, None, None, None


def bn_bcast_act(y0, t, bn, K, slope=0.0):
    """act(bn(y0 + t[group])) through BnBcastActFn for an nn.BatchNorm1d module `bn` (rows layout, K rows per group)."""
    meta = {"adt": _adt(), "training": bn.training, "eps": bn.eps, "momentum": bn.momentum, "slope": slope, "K": K,
            "grad": torch.is_grad_enabled()}
    return BnBcastActFn.apply(y0, t, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, meta)


def bn_bcast_supported(x, C, K):
    return x.is_cuda and C % 8 == 0 and 8 <= C <= 1024 and 1 <= K <= 255


class ExpandRowsFn(torch.autograd.Function):
    """token (1,1,C) -> (B,N,C); backward = column sum over the B*N rows with our two-stage kernel."""

    @staticmethod
    def forward(ctx, token, B, N, dtype):
        ctx.tdt = token.dtype
        return weight_cache.get(token, dtype).expand(B, N, -1)      # (the optimizer's shadow where there is one: no cast launch)

    @staticmethod
    def backward(ctx, g):
        with torch.autocast("cuda", enabled=False):
            C = g.shape[-1]
            g2 = g.reshape(-1, C).contiguous()
            return colsum(g2, g2.dtype).view(1, 1, C).to(ctx.tdt), None, None, None


class _RankLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        with torch.autocast("cuda", enabled=False):
            B, M = pred.shape
            p = pred.detach().float().contiguous()
            t = target.detach().float().contiguous()
            out = torch.empty(B, 2, dtype=torch.float32, device=pred.device)
            dp = torch.empty(B, M, dtype=torch.float32, device=pred.device)
            _launch("gm3d_rank_loss", {"B": B, "M": M}, lib.gm3d_rank_loss, _ptr(p), _ptr(t), B, M, _ptr(out), _ptr(dp), _stream())
            tot = out.sum(dim=0)                                 # B rows: single-block reduction
            ctx.save_for_backward(dp, tot)
            ctx.pdt = pred.dtype
            return tot[0] / tot[1]

    @staticmethod
    def backward(ctx, g):
        dp, tot = ctx.saved_tensors
        return (dp * (g / tot[1])).to(ctx.pdt), None


class _RankLossTail(torch.autograd.Function):
    """loss_learn on the last M columns of the full (B,L) f32 prediction: kernel + fixed-order finish (2 launches), backward ONE launch
    that writes the gradient of the full prediction (zeros in front) -- no slice, clone, sum, division or product on the PyTorch side."""

    @staticmethod
    def forward(ctx, pred_full, M, target):
        with torch.autocast("cuda", enabled=False):
            B, L = pred_full.shape
            dev = pred_full.device
            t = target.detach().float().contiguous()
            out = torch.empty(B, 2, dtype=torch.float32, device=dev)
            dp = torch.empty(B, M, dtype=torch.float32, device=dev)
            tot = torch.empty(2, dtype=torch.float32, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            _launch("gm3d_rank_loss_tail", {"B": B, "M": M}, lib.gm3d_rank_loss_tail, pred_full.data_ptr() + 4 * (L - M), pred_full.stride(0),
                    _ptr(t), B, M, _ptr(out), _ptr(dp), _ptr(tot), _ptr(loss), _stream())
            ctx.save_for_backward(dp, tot)
            ctx.dims = (B, M, L)
            return loss

    @staticmethod
    def backward(ctx, g):
        dp, tot = ctx.saved_tensors
        B, M, L = ctx.dims
        g = g.detach().reshape(1).float().contiguous()
        dfull = torch.empty(B, L, dtype=torch.float32, device=dp.device)
        _launch("gm3d_rank_loss_tail_bwd", {"B": B, "M": M, "L": L}, lib.gm3d_rank_loss_tail_bwd, _ptr(dp), _ptr(g), _ptr(tot), B, M, L,
                _ptr(dfull), _stream())
        return dfull, None, None


def rank_loss_tail(pred_full, M, target):
    """rank_loss(pred_full[:, -M:], target) with the slice folded in; pred_full (B,L) f32, unit inner stride."""
    if pred_full.dtype == torch.float32 and pred_full.dim() == 2 and pred_full.stride(1) == 1 and 1 <= M <= min(64, pred_full.shape[1]):
        return _RankLossTail.apply(pred_full, M, target)
    return _RankLoss.apply(pred_full[:, -M:], target)


class PatchChamferLossFn(torch.autograd.Function):
    """forward_loss of the north-star model on the masked patches in one pass (csrc/chamfer.hip patch_loss_*): pred (B,M,96) -- a
    batch-strided view of the decoder head's (B,L,96) output is taken as it is --, target (B,T,32,3) f32, mask_ids (B,M) int64
    -> (Chamfer_mean (), matrix (B,M) f32).  Only Chamfer_mean carries a gradient (the engine detaches `matrix`: it is the teacher's
    ranking target, P/engine_pretrain.py:156-171)."""

    @staticmethod
    def forward(ctx, pred, target, mask_ids, full=False):
        # full=True: pred is the whole (B,L,96) prediction, contiguous; its last M patches enter the loss and the backward returns the
        # gradient of the whole tensor (zeros for the visible patches) -- the [:, -M:] slice and its zero-filling backward folded in
        ctx.set_materialize_grads(False)
        ctx.lead = 0
        if full:
            ctx.lead = pred.shape[1] - mask_ids.shape[1]
            pred = pred.detach()[:, ctx.lead:]
        B, M, _ = pred.shape
        T = target.shape[1]
        dev = pred.device
        matrix = torch.empty(B, M, dtype=torch.float32, device=dev)
        i1 = torch.empty(B * M, 32, dtype=torch.int32, device=dev)
        i2 = torch.empty(B * M, 32, dtype=torch.int32, device=dev)
        mean = torch.empty((), dtype=torch.float32, device=dev)
        _launch("gm3d_patch_chamfer_loss_fwd", {"B": B, "M": M, "dtype": str(pred.dtype)}, lib.gm3d_patch_chamfer_loss_fwd, _ptr(pred),
                pred.stride(0), _ptr(target), _ptr(mask_ids), mask_ids.stride(0), B, T, M, _ptr(matrix), _ptr(i1), _ptr(i2), _ptr(mean),
                _DT[pred.dtype], _stream())
        ctx.save_for_backward(pred, target, mask_ids, i1, i2)
        ctx.mark_non_differentiable(matrix)
        return mean, matrix

    @staticmethod
    def backward(ctx, g, _gm):
        pred, target, mask_ids, i1, i2 = ctx.saved_tensors
        B, M, _ = pred.shape
        if g is None:
            return None, None, None, None
        g = g.detach().reshape(1).float().contiguous()
        dpred = torch.empty(B, ctx.lead + M, 96, dtype=pred.dtype, device=pred.device)
        _launch("gm3d_patch_chamfer_loss_bwd", {"B": B, "M": M, "dtype": str(pred.dtype)}, lib.gm3d_patch_chamfer_loss_bwd_full, _ptr(pred),
                pred.stride(0), _ptr(target), _ptr(mask_ids), mask_ids.stride(0), _ptr(i1), _ptr(i2), _ptr(g), B, target.shape[1], M,
                ctx.lead, _ptr(dpred), _DT[pred.dtype], _stream())
        return dpred, None, None, None


FUSED_PATCH_LOSS = True


def patch_chamfer_loss_supported(pred, target, mask_ids):
    return (FUSED_PATCH_LOSS and pred.is_cuda and pred.dim() == 3 and pred.shape[-1] == 96 and pred.stride(2) == 1 and pred.stride(1) == 96
            and pred.dtype in _DT and target.dtype == torch.float32 and target.is_contiguous() and tuple(target.shape[2:]) == (32, 3)
            and mask_ids is not None and mask_ids.dtype == torch.int64 and mask_ids.stride(1) == 1
            and mask_ids.shape == pred.shape[:2] and not target.requires_grad)


def rank_loss(pred, target):
    return _RankLoss.apply(pred, target)


class TokenAssembleFn(torch.autograd.Function):
    """tokens, pos (B,L,C), order (B,L) int64 = [visible ids | masked ids], V -> x_vis, pos_vis (B,V,C), pos_full (B,L,C):
    the boolean-mask gathers + concat of P/models_mae_learn_loss.py:298-300,649-658 as one launch; backward one launch, exact
    (order is a permutation: no atomics)."""

    @staticmethod
    def forward(ctx, tokens, pos, order, V):
        B, L, C = tokens.shape
        dt = tokens.dtype
        tokens = tokens.contiguous()
        pos = pos.to(dt).contiguous()
        order = order.contiguous()
        x_vis = torch.empty(B, V, C, dtype=dt, device=tokens.device)
        pos_vis = torch.empty(B, V, C, dtype=dt, device=tokens.device)
        pos_full = torch.empty(B, L, C, dtype=dt, device=tokens.device)
        _launch("gm3d_token_assemble_fwd", {"B": B, "L": L, "C": C}, lib.gm3d_token_assemble_fwd, _ptr(tokens), _ptr(pos), _ptr(order),
                B, L, V, C, _ptr(x_vis), _ptr(pos_vis), _ptr(pos_full), _DT[dt], _stream())
        ctx.save_for_backward(order)
        ctx.dims, ctx.dt, ctx.pdt = (B, L, V, C), dt, pos.dtype
        return x_vis, pos_vis, pos_full

    @staticmethod
    def backward(ctx, dx_vis, dpos_vis, dpos_full):
        (order,) = ctx.saved_tensors
        B, L, V, C = ctx.dims
        dt = ctx.dt
        g = [None if t is None else t.to(dt).contiguous() for t in (dx_vis, dpos_vis, dpos_full)]
        dtokens = torch.empty(B, L, C, dtype=dt, device=order.device)
        dpos = torch.empty(B, L, C, dtype=dt, device=order.device)
        _launch("gm3d_token_assemble_bwd", {"B": B, "L": L, "C": C}, lib.gm3d_token_assemble_bwd, _ptr(g[0]), _ptr(g[1]), _ptr(g[2]),
                _ptr(order), B, L, V, C, _ptr(dtokens), _ptr(dpos), _DT[dt], _stream())
        return dtokens, dpos, None, None


class PosAssembleFn(torch.autograd.Function):
    """TokenAssembleFn without the token half (the tokens were embedded for the visible groups only): pos (B,L,C), order ->
    pos_vis (B,V,C), pos_full (B,L,C)."""

    @staticmethod
    def forward(ctx, pos, order, V, dt):
        B, L, C = pos.shape
        pos = pos.to(dt).contiguous()
        order = order.contiguous()
        pos_vis = torch.empty(B, V, C, dtype=dt, device=pos.device)
        pos_full = torch.empty(B, L, C, dtype=dt, device=pos.device)
        _launch("gm3d_token_assemble_fwd", {"B": B, "L": L, "C": C}, lib.gm3d_token_assemble_fwd, None, _ptr(pos), _ptr(order),
                B, L, V, C, None, _ptr(pos_vis), _ptr(pos_full), _DT[dt], _stream())
        ctx.save_for_backward(order)
        ctx.dims, ctx.dt = (B, L, V, C), dt
        return pos_vis, pos_full

    @staticmethod
    def backward(ctx, dpos_vis, dpos_full):
        (order,) = ctx.saved_tensors
        B, L, V, C = ctx.dims
        dt = ctx.dt
        g = [None if t is None else t.to(dt).contiguous() for t in (dpos_vis, dpos_full)]
        dpos = torch.empty(B, L, C, dtype=dt, device=order.device)
        _launch("gm3d_token_assemble_bwd", {"B": B, "L": L, "C": C}, lib.gm3d_token_assemble_bwd, None, _ptr(g[0]), _ptr(g[1]),
                _ptr(order), B, L, V, C, None, _ptr(dpos), _DT[dt], _stream())
        return dpos, None, None, None


def _order_of(vis_ids, mask_ids):
    L = vis_ids.shape[1] + mask_ids.shape[1]
    if (vis_ids.stride(0) == L and vis_ids.stride(1) == 1 and mask_ids.shape[1] and mask_ids.stride(0) == L and
            mask_ids.data_ptr() == vis_ids.data_ptr() + vis_ids.shape[1] * vis_ids.element_size()):
        return torch.as_strided(vis_ids, (vis_ids.shape[0], L), (L, 1))      # the two halves of one (B,L) buffer
    if mask_ids.shape[1] == 0:
        return vis_ids
    return torch.cat([vis_ids, mask_ids], dim=1)


def pos_assemble(pos, vis_ids, mask_ids, dt=None):
    return PosAssembleFn.apply(pos, _order_of(vis_ids, mask_ids), vis_ids.shape[1], dt or _adt())


def token_assemble(tokens, pos, vis_ids, mask_ids, order=None):
    if order is None:
        L = vis_ids.shape[1] + mask_ids.shape[1]
        if (vis_ids.stride(0) == L and vis_ids.stride(1) == 1 and mask_ids.shape[1] and mask_ids.stride(0) == L and
                mask_ids.data_ptr() == vis_ids.data_ptr() + vis_ids.shape[1] * vis_ids.element_size()):
            order = torch.as_strided(vis_ids, (vis_ids.shape[0], L), (L, 1))      # the two halves of one (B,L) buffer
        elif mask_ids.shape[1] == 0:
            order = vis_ids
        else:
            order = torch.cat([vis_ids, mask_ids], dim=1)
    return TokenAssembleFn.apply(tokens, pos, order, vis_ids.shape[1])
