"""Mini-PointNet token embed (Encoder.forward, Point-MAE_SA3D/models_mae_learn_loss.py:868-899) as ONE autograd
node: the wide products on our own MFMA kernels (gemm.mm / gemm.mm_nn: no library GEMM in bf16 mode since round 3) with every
pass between them a hand-written streaming kernel (gm3d_amd/csrc/embed.hip), forward and backward.

    x (rows,3) --[K=3 conv + BN1 + ReLU on the fly]--> a1 (rows,128) --GEMM--> f (rows,256)
      fg = max_k f ; t = fg @ W3[:, :256]^T + b3 (per group) ; y0 = f @ W3[:, 256:]^T
      a2 = ReLU(BN2(y0 + t[group])) --GEMM--> z (rows,384) ; tokens = max_k z + b4

Identities used (exact up to fp32 rounding order): concat([global,local]) @ W = local @ W_l + global @ W_g;
max_k(z) + b = max_k(z + b); the batch statistics of the K=3 layer follow from the 3x3 input moments; a bias in
front of a BatchNorm has zero gradient.  Train mode updates the BatchNorm running statistics exactly like
nn.BatchNorm1d (momentum 0.1, unbiased running variance, num_batches_tracked += 1).
"""
import torch

from ._capi import lib
from .fused import weight_cache, LNC  # noqa: F401
from . import gemm
from .ops import _launch, _ptr, _stream, _DT

SPLITK = 64  # row chunks of the weight-gradient GEMMs (K = 262,144 rows would otherwise run on a handful of CUs)


def _finish(partial, nrows, ncols, pitch=None):
    out = torch.empty(ncols, dtype=torch.float32, device=partial.device)
    _launch("gm3d_colsum_finish", {"rows": nrows, "cols": ncols}, lib.gm3d_colsum_finish, _ptr(partial), nrows,
            pitch or ncols, ncols, _ptr(out), 0, _stream())
    return out


def colsum(m, adt):
    """column sums of a contiguous (R,C) matrix -> (C,) f32 (two deterministic stages)."""
    R, C = m.shape
    nrows = lib.gm3d_embed_partial_rows(2, R, C)
    partial = torch.empty(nrows, C, dtype=torch.float32, device=m.device)
    _launch("gm3d_colsum_partial", {"R": R, "C": C, "dtype": str(m.dtype)}, lib.gm3d_colsum_partial, _ptr(m), R, C,
            _ptr(partial), _DT[m.dtype], _stream())
    return _finish(partial, nrows, C)


def splitk_wgrad(dy, x):
    """dW (N,K) f32 = dy (R,N)^T @ x (R,K) with R split into SPLITK batched chunks + a deterministic sum."""
    R, N = dy.shape
    K = x.shape[1]
    from . import gemm
    if dy.is_contiguous() and x.is_contiguous() and gemm.wgrad_supported(dy.unsqueeze(0), x.unsqueeze(0)):
        return gemm.wgrad_nt(dy.unsqueeze(0), x.unsqueeze(0))[0]      # hand-written NT kernel (row split chosen by shape)
    if (N % 128 and N < 128 and dy.is_contiguous() and x.is_contiguous() and dy.dtype == torch.bfloat16
            and gemm.wgrad_supported(dy.new_empty(1, R, 128), x.unsqueeze(0))):
        # a narrow output (the 96-wide reconstruction head): zero-padded to one 128-column tile for the NT kernel (the library
        # runs this shape on 32x32 tiles: 88 us per step)
        pad = torch.zeros(R, 128, dtype=dy.dtype, device=dy.device)
        pad[:, :N] = dy
        return gemm.wgrad_nt(pad.unsqueeze(0), x.unsqueeze(0))[0][:N].contiguous()
    S = SPLITK if R % SPLITK == 0 and R >= 64 * SPLITK else 1
    a = dy.view(S, R // S, N).transpose(1, 2)
    b = x.view(S, R // S, K)
    if dy.dtype == torch.float32:
        part = torch.bmm(a, b)
    else:
        try:
            part = torch.bmm(a, b, out_dtype=torch.float32)
        except TypeError:
            part = torch.bmm(a, b).float()
    if S == 1:
        return part[0]
    from . import fused
    if not fused.SUM_FEW_ROWS:
        return _finish(part.view(S, N * K), S, N * K).view(N, K)
    out = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    _launch("gm3d_sum_few_rows", {"rows": S, "cols": N * K}, lib.gm3d_sum_few_rows, _ptr(part), 1, S, N * K, _ptr(out), _stream())
    return out


def _c32(p):
    """detached, contiguous fp32 view of a parameter/buffer (no launch when it already is one)."""
    p = p.detach()
    return p if (p.dtype == torch.float32 and p.is_contiguous()) else p.float().contiguous()


class EmbedFn(torch.autograd.Function):
    """args: nb (B,G,K,3) f32, meta, w1,b1,g1,be1, w2,b2, w3,b3,g2,be2, w4,b4, then the BatchNorm buffers
    rm1,rv1,nbt1, rm2,rv2,nbt2 (updated in place in train mode).  Returns tokens (B,G,C4) in meta['adt'].

    meta['vis_ids'] (B,V) int64 (optional; a strided view of a (B,L) buffer is fine): only these groups' tokens are wanted --
    the student keeps 25 of 64 (x_vis = tokens[~mask], P/models_mae_learn_loss.py:298) and nothing reads the other 39.  Every
    layer up to and including the second BatchNorm's statistics still runs over all rows (the statistics, and the running
    buffers, are those of the full batch); BN-apply + ReLU, the last conv and its max-pool run on the selected groups only,
    and so do that conv's input- and weight-gradient GEMMs (the dropped rows' gradient is exactly zero).  Returns (B,V,C4)."""

    @staticmethod
    def forward(ctx, nb, meta, w1, b1, g1, be1, w2, b2, w3, b3, g2, be2, w4, b4, rm1, rv1, nbt1, rm2, rv2, nbt2):
        with torch.autocast("cuda", enabled=False):
            return EmbedFn._forward(ctx, nb, meta, w1, b1, g1, be1, w2, b2, w3, b3, g2, be2, w4, b4, rm1, rv1, nbt1,
                                    rm2, rv2, nbt2)

    @staticmethod
    def _forward(ctx, nb, meta, w1, b1, g1, be1, w2, b2, w3, b3, g2, be2, w4, b4, rm1, rv1, nbt1, rm2, rv2, nbt2):
        adt, training, eps, mom = meta["adt"], meta["training"], meta["eps"], meta["momentum"]
        dt_id = _DT[adt]
        B, G, K, _ = nb.shape
        BG, R = B * G, B * G * K
        dev = nb.device
        x = nb.reshape(R, 3).float().contiguous()
        C1, C2, C3, C4 = w1.shape[0], w2.shape[0], w3.shape[0], w4.shape[0]
        W1 = w1.reshape(C1, 3)
        f64 = torch.float64
        # ---- layer 1 statistics (analytic in the input moments), folded into the conv by one tiny kernel ----
        f32 = dict(dtype=torch.float32, device=dev)
        mom9 = mcov = xmean = None
        if training:
            nrows = lib.gm3d_embed_partial_rows(0, R, 0)
            part = torch.empty(nrows, 9, dtype=f64, device=dev)
            _launch("gm3d_moments3", {"R": R}, lib.gm3d_moments3, _ptr(x), R, _ptr(part), _stream())
            mom9 = torch.empty(9, dtype=f64, device=dev)
            _launch("gm3d_colsum_finish_f64", {"rows": nrows, "cols": 9}, lib.gm3d_colsum_finish_f64, _ptr(part), nrows, 9, 9,
                    _ptr(mom9), _stream())
            mcov = torch.empty(12, dtype=f64, device=dev)
            xmean = torch.empty(3, **f32)
        w1c, b1c, g1c, be1c = _c32(W1), _c32(b1), _c32(g1), _c32(be1)
        wf, bf = torch.empty(C1, 3, **f32), torch.empty(C1, **f32)
        mean1, rstd1 = torch.empty(C1, **f32), torch.empty(C1, **f32)
        _launch("gm3d_pn1_finalize", {"C": C1}, lib.gm3d_pn1_finalize, _ptr(mom9), float(R), _ptr(w1c), _ptr(b1c), _ptr(g1c),
                _ptr(be1c), eps, mom, _ptr(rm1), _ptr(rv1), _ptr(nbt1), _ptr(wf), _ptr(bf), _ptr(mean1), _ptr(rstd1),
                _ptr(mcov), _ptr(xmean), C1, int(training), _stream())
        a1 = torch.empty(R, C1, dtype=adt, device=dev)
        _launch("gm3d_pn_layer1_fwd", {"R": R, "C": C1, "dtype": str(adt)}, lib.gm3d_pn_layer1_fwd, _ptr(x), _ptr(wf),
                _ptr(bf), _ptr(a1), R, C1, dt_id, _stream())
        # ---- conv2 + max-pool ----
        W2 = weight_cache.get(w2, adt).reshape(C2, C1)
        pool_fused = gemm.FUSE_POOL and ((K == 32 and gemm.supported(a1, W2)) or (K == 16 and gemm.pool16_supported(a1, W2)))
        if pool_fused:      # conv2 + bias + max-pool in one launch: f is written for conv3, (fg, arg1) come from the same tile
            f, fg, arg1 = gemm.linear_pool(a1, W2, _c32(b2), bias_after_pool=False, want_rows=True, group_rows=K)
        else:
            f = gemm.mm(a1, W2, _c32(b2)) if adt == torch.bfloat16 else torch.addmm(weight_cache.get(b2, adt), a1, W2.t())
            fg = torch.empty(BG, C2, dtype=adt, device=dev)
            arg1 = torch.empty(BG, C2, dtype=torch.uint8, device=dev)
            _launch("gm3d_group_max_fwd", {"G": BG, "K": K, "C": C2, "dtype": str(adt)}, lib.gm3d_group_max_fwd, _ptr(f), None,
                    _ptr(fg), _ptr(arg1), BG, K, C2, dt_id, _stream())
        # ---- conv3 on [global | local] ----
        W3 = weight_cache.get(w3, adt).reshape(C3, 2 * C2)
        W3g, W3l = W3[:, :C2], W3[:, C2:]
        t = gemm.mm(fg, W3g, _c32(b3))
        vis = meta.get("vis_ids")
        # the product with the BatchNorm behind it in its epilogue (csrc/gemm_ws.hip EPI 4 / 5): train mode -> its statistics come
        # out of the product's launch (no second pass over the (rows, 512) tensor); eval mode (the EMA teacher) -> BN + ReLU applied
        # there, the product itself never reaches HBM
        fuse_bn = K in (16, 32) and gemm.ws_bn_supported(f, W3l, t, group_rows=K)
        y0 = st = None
        if training:
            # (groups of 16 rows: the statistics epilogue measured SLOWER than product + pass -- 711 vs 355 + 245 us at 1,048,576 rows,
            #  tools/m2ae_gemm_shapes.py -- so only the eval-mode apply epilogue is used there: 654 vs 355 + 388)
            if fuse_bn and (K == 32 or gemm.WS_BN_STATS16):
                y0, part = gemm.linear_ws_bn_stats(f, W3l, t, group_rows=K)
                st = _finish(part, part.shape[0], 2 * C3)
            else:
                y0 = gemm.mm(f, W3l)
                nrows = lib.gm3d_embed_partial_rows(1, BG, C3)
                part = torch.empty(nrows, 2 * C3, dtype=torch.float32, device=dev)
                _launch("gm3d_bn_bcast_stats", {"G": BG, "K": K, "C": C3, "dtype": str(adt)}, lib.gm3d_bn_bcast_stats, _ptr(y0),
                        _ptr(t), BG, K, C3, _ptr(part), dt_id, _stream())
                st = _finish(part, nrows, 2 * C3)
        g2c, be2c = _c32(g2), _c32(be2)
        scale2, shift2 = torch.empty(C3, **f32), torch.empty(C3, **f32)
        mean2, rstd2 = torch.empty(C3, **f32), torch.empty(C3, **f32)
        _launch("gm3d_bn_finalize", {"C": C3}, lib.gm3d_bn_finalize, _ptr(st), float(R), _ptr(g2c), _ptr(be2c), eps, mom,
                _ptr(rm2), _ptr(rv2), _ptr(nbt2), _ptr(scale2), _ptr(shift2), _ptr(mean2), _ptr(rstd2), C3, int(training),
                _stream())
        sel = inv = None
        BGs, Gs = BG, G                       # groups that go through conv4
        if vis is not None:
            if vis.dim() != 2 or vis.shape[0] != B or vis.dtype != torch.int64 or vis.stride(1) != 1 or vis.shape[1] > G:
                raise ValueError("vis_ids must be (B,V) int64 with unit inner stride, V <= G")
            Gs = vis.shape[1]
            BGs = B * Gs
            sel = torch.empty(BGs, dtype=torch.int32, device=dev)
            inv = torch.empty(BG, dtype=torch.int32, device=dev)
            _launch("gm3d_group_select_maps", {"B": B, "V": Gs, "G": G}, lib.gm3d_group_select_maps, _ptr(vis), vis.stride(0), B,
                    Gs, G, _ptr(sel), _ptr(inv), _stream())
        if not training and fuse_bn and vis is None:
            a2 = gemm.linear_ws_bn_apply(f, W3l, t, scale2, shift2, group_rows=K)
        else:
            if y0 is None:
                y0 = gemm.mm(f, W3l)
            a2 = torch.empty(BGs * K, C3, dtype=adt, device=dev)
            _launch("gm3d_bn_bcast_apply_relu", {"G": BGs, "K": K, "C": C3, "dtype": str(adt)}, lib.gm3d_bn_bcast_apply_relu_sel,
                    _ptr(y0), _ptr(t), _ptr(scale2), _ptr(shift2), _ptr(a2), _ptr(sel), BGs, K, C3, 0.0, dt_id, _stream())
        # ---- conv4 + max-pool ----
        W4 = weight_cache.get(w4, adt).reshape(C4, C3)
        b4f = _c32(b4)
        if gemm.FUSE_POOL and ((K == 32 and gemm.supported(a2, W4)) or (K == 16 and gemm.pool16_supported(a2, W4))):
            # conv4 + max-pool + bias: the (rows, 384) product never reaches HBM (the backward needs only arg2)
            _, tok, arg2 = gemm.linear_pool(a2, W4, b4f, bias_after_pool=True, want_rows=False, group_rows=K)
        else:
            z = gemm.mm(a2, W4) if adt == torch.bfloat16 else a2 @ W4.t()     # groups of k != 32 points (Point-M2AE level 0: k = 16)
            tok = torch.empty(BGs, C4, dtype=adt, device=dev)
            arg2 = torch.empty(BGs, C4, dtype=torch.uint8, device=dev)
            _launch("gm3d_group_max_fwd", {"G": BGs, "K": K, "C": C4, "dtype": str(adt)}, lib.gm3d_group_max_fwd, _ptr(z),
                    _ptr(b4f), _ptr(tok), _ptr(arg2), BGs, K, C4, dt_id, _stream())
        if meta["grad"] and any(ctx.needs_input_grad):
            if not training:
                raise NotImplementedError("EmbedFn backward is implemented for train-mode BatchNorm only")
            ctx.save_for_backward(x, a1, f, fg, arg1, y0, t, a2, arg2, w1, b1, g1, w2, w3, g2, w4, mcov, xmean,
                                  mean1, rstd1, mean2, rstd2, scale2, shift2)
            ctx.meta, ctx.dims = meta, (B, G, K, C1, C2, C3, C4)
            ctx.sel, ctx.inv, ctx.Gs = sel, inv, Gs
        return tok.view(B, Gs, C4)

    @staticmethod
    def backward(ctx, dtok):
        with torch.autocast("cuda", enabled=False):
            return EmbedFn._backward(ctx, dtok)

    @staticmethod
    def _backward(ctx, dtok):
        (x, a1, f, fg, arg1, y0, t, a2, arg2, w1, b1, g1, w2, w3, g2, w4, mcov, xmean, mean1, rstd1, mean2, rstd2, scale2,
         shift2) = ctx.saved_tensors
        adt = ctx.meta["adt"]
        dt_id = _DT[adt]
        B, G, K, C1, C2, C3, C4 = ctx.dims
        BG, R = B * G, B * G * K
        dev = dtok.device
        f64 = torch.float64
        sel, inv, BGs = ctx.sel, ctx.inv, B * ctx.Gs         # conv4 saw BGs groups (all of them when sel is None)
        dtok = dtok.reshape(BGs, C4).to(adt).contiguous()
        db4 = colsum(dtok, adt)
        # conv4
        dz = torch.empty(BGs * K, C4, dtype=adt, device=dev)
        _launch("gm3d_group_max_bwd", {"G": BGs, "K": K, "C": C4, "dtype": str(adt)}, lib.gm3d_group_max_bwd, _ptr(dtok),
                _ptr(arg2), _ptr(dz), BGs, K, C4, dt_id, _stream())
        W4 = weight_cache.get(w4, adt).reshape(C4, C3)
        da2 = gemm.mm_nn(dz, W4)
        # the four weight-gradient products of this node: ONE launch at the end (gemm.wgrad_nt_multi: together they fill the chip with
        # row splits in proportion to their rows) instead of four launches that each cut themselves into 64 slabs
        later = []
        multi = gemm.MULTI_WGRAD and adt == torch.bfloat16 and gemm.ENABLED

        def wgrad(dy_, x_):
            if multi and dy_.is_contiguous() and x_.is_contiguous() and gemm.wgrad_multi_ok(dy_.unsqueeze(0), x_.unsqueeze(0), None):
                out = torch.empty(1, dy_.shape[1], x_.shape[1], dtype=torch.float32, device=dev)
                later.append((dy_.unsqueeze(0), x_.unsqueeze(0), out))
                return out[0]
            return splitk_wgrad(dy_, x_)
        dW4 = wgrad(dz, a2)
        # BN2 + ReLU: the sums run over the rows that carry a gradient, dy is written for every row
        nrows = lib.gm3d_embed_partial_rows(1, BGs, C3)
        part = torch.empty(nrows, 2 * C3, dtype=torch.float32, device=dev)
        _launch("gm3d_bn_bcast_bwd_stats", {"G": BGs, "K": K, "C": C3, "dtype": str(adt)}, lib.gm3d_bn_bcast_bwd_stats_sel,
                _ptr(da2), _ptr(y0), _ptr(t), _ptr(scale2), _ptr(shift2), _ptr(mean2), _ptr(rstd2), _ptr(sel), BGs, K, C3,
                _ptr(part), 0.0, dt_id, _stream())
        s12 = _finish(part, nrows, 2 * C3)
        s1, s2 = s12[:C3], s12[C3:]
        dy = torch.empty(R, C3, dtype=adt, device=dev)
        dt = torch.empty(BG, C3, dtype=torch.float32, device=dev)
        _launch("gm3d_bn_bcast_bwd_apply", {"G": BG, "K": K, "C": C3, "dtype": str(adt)}, lib.gm3d_bn_bcast_bwd_apply_sel,
                _ptr(da2), _ptr(y0), _ptr(t), _ptr(scale2), _ptr(shift2), _ptr(mean2), _ptr(rstd2), _ptr(s1), _ptr(s2),
                _ptr(dy), _ptr(dt), _ptr(inv), BG, K, C3, 0.0, dt_id, _stream())
        dg2, dbe2 = s2, s1
        # conv3: local part on rows, global part per group
        W3 = weight_cache.get(w3, adt).reshape(C3, 2 * C2)
        W3g, W3l = W3[:, :C2], W3[:, C2:]
        bf16 = adt == torch.bfloat16 and gemm.ENABLED
        if bf16:            # one transposing launch for both halves: rows [0, C2) of W3^T multiply the per-group term, the rest the rows
            W3T = gemm.transposed(W3)
        df = gemm.mm(dy, W3T[C2:]) if bf16 else dy @ W3l
        dW3l = wgrad(dy, f)
        db3 = colsum(dt, torch.float32)
        dta = dt.to(adt)
        dW3g = wgrad(dta, fg)
        dfg = gemm.mm(dta, W3T[:C2]) if bf16 else dta @ W3g
        # max-pool branch joins df; bias grad of conv2
        nrows = lib.gm3d_embed_partial_rows(1, BG, C2)
        part = torch.empty(nrows, C2, dtype=torch.float32, device=dev)
        _launch("gm3d_group_scatter_add", {"G": BG, "K": K, "C": C2, "dtype": str(adt)}, lib.gm3d_group_scatter_add, _ptr(df),
                _ptr(dfg), _ptr(arg1), BG, K, C2, _ptr(part), dt_id, _stream())
        db2 = _finish(part, nrows, C2)
        # conv2
        W2 = weight_cache.get(w2, adt).reshape(C2, C1)
        da1 = gemm.mm_nn(df, W2)
        dW2 = wgrad(df, a1).reshape(w2.shape)
        # layer 1: BN1 + conv(K=3), reductions only
        W1 = _c32(w1.reshape(C1, 3))
        nrows = lib.gm3d_embed_partial_rows(3, R, C1)
        part = torch.empty(nrows, 5 * C1, dtype=f64, device=dev)
        b1f, g1f = _c32(b1), _c32(g1)
        _launch("gm3d_pn_layer1_bwd_stats", {"R": R, "C": C1, "dtype": str(adt)}, lib.gm3d_pn_layer1_bwd_stats, _ptr(da1),
                _ptr(a1), _ptr(x), _ptr(W1), _ptr(b1f), _ptr(mean1), _ptr(rstd1), _ptr(xmean), R, C1, _ptr(part), dt_id,
                _stream())
        q = torch.empty(5 * C1, dtype=f64, device=dev)
        _launch("gm3d_colsum_finish_f64", {"rows": nrows, "cols": 5 * C1}, lib.gm3d_colsum_finish_f64, _ptr(part), nrows,
                5 * C1, 5 * C1, _ptr(q), _stream())
        # q = [sum g1, sum g1*hhat, sum g1*(x_j - mean_j)];  dh0 = k*(g1 - t1/R - hhat*t2/R), k = gamma*rstd;
        # dW1[c,j] = sum_r dh0*(x_j - m_j) (sum_r dh0 = 0) and sum_r hhat*(x_j - m_j) = rstd * R * (W1 cov)_j
        dW1 = torch.empty(C1, 3, dtype=torch.float32, device=dev)
        dg1, dbe1 = torch.empty(C1, dtype=torch.float32, device=dev), torch.empty(C1, dtype=torch.float32, device=dev)
        _launch("gm3d_pn1_bwd_finalize", {"C": C1}, lib.gm3d_pn1_bwd_finalize, _ptr(q), _ptr(mcov), _ptr(W1), _ptr(g1f),
                _ptr(rstd1), _ptr(dW1), _ptr(dg1), _ptr(dbe1), C1, _stream())
        db1 = torch.zeros_like(b1)                                       # bias in front of BatchNorm: exactly zero
        if len(later) > 1:
            gemm.wgrad_nt_multi(later)
        elif later:
            gemm.wgrad_nt(later[0][0], later[0][1], later[0][2])
        dW3 = torch.cat([dW3g, dW3l], dim=1).reshape(w3.shape)
        return (None, None, dW1.reshape(w1.shape), db1, dg1, dbe1, dW2, db2, dW3, db3, dg2, dbe2,
                dW4.reshape(w4.shape), db4, None, None, None, None, None, None)


def run_embed(enc, point_groups, vis_ids=None):
    """enc: the Encoder module (parameters in the reference's Conv1d/BatchNorm1d layout).  vis_ids (B,V) int64: embed only these
    groups' tokens -> (B,V,C) (see EmbedFn)."""
    c0, bn0, _, c1 = enc.first_conv
    c2, bn1, _, c3 = enc.second_conv
    adt = torch.bfloat16 if (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16) \
        else torch.float32
    meta = {"adt": adt, "training": enc.training, "eps": bn0.eps, "momentum": bn0.momentum,
            "grad": torch.is_grad_enabled(), "vis_ids": vis_ids}
    return EmbedFn.apply(point_groups, meta, c0.weight, c0.bias, bn0.weight, bn0.bias, c1.weight, c1.bias, c2.weight,
                         c2.bias, bn1.weight, bn1.bias, c3.weight, c3.bias, bn0.running_mean, bn0.running_var,
                         bn0.num_batches_tracked, bn1.running_mean, bn1.running_var, bn1.num_batches_tracked)
