"""Block stack of the hierarchical (Point-M2AE) encoder / decoder as ONE autograd node, and the visible-first token order the
student's pass runs it in.

Beneath: the pre-norm transformer blocks of Point-M2AE's H_Encoder / H_Decoder (the reference ships only the
hyper-parameters, Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99: widths 96 / 192 / 384, 5 + 5 + 5 encoder and 1 + 1 decoder
blocks, 6 heads, local radii 0.32 / 0.64 / 1.28, mask ratio 0.8), driven like Point-MAE's stacks (the position is re-added in
front of every block, P/models_mae_learn_loss.py:914-917):

    for block in blocks:   x = block(x + pos)      # x += dp(attn(LN1(x), mask));  x += dp(mlp(LN2(x)))

MaskedStackFn is to this model what fused.TransformerStackFn is to the north-star model, on the any-width kernels: residual sums
inside the LayerNorm launches (gm3d_add_ln_fwd / _bwd_acc), bias + GELU as one pass, masked flash attention
(gm3d_attention_masked_*), every product on our own MFMA kernels (gemm.mm), and -- what the per-op nodes cannot do -- the
stack-wide work batched: the four weight gradients of all blocks in ONE launch (gemm.wgrad_nt_multi, written straight into the
flat gradient buffer), every LayerNorm / GELU column-sum finish in ONE launch each, the four transposed weight shadows in ONE
launch, the positional embedding's gradient accumulated inside the LayerNorm backward.

Visible-first order: gm3d_partition_visible + CompactFn / MergeFn move a level's visible tokens to the front of every cloud (cut to a
static bound where the mask generator fixes the count), so that the attention kernels skip the filler tiles and -- with a bound
below the token count -- every row pass shrinks.  Results for visible tokens are those of the in-place form (row-local layers;
attention sees the same allowed keys in the same relative order); filler rows are never read back.
"""
import torch

from ._capi import lib
from . import fused, gemm
from .embed import _c32
from .fused import PER_BLOCK, block_params, weight_cache  # noqa: F401
from .ops import _DT, _launch, _ptr, _stream


# ----------------------------------------------------------------------------- kernel wrappers
def _add_ln_fwd(x, y, ybias, rowscale, rps, z, gamma, beta, eps, adt, want_s, h_out=None):
    """-> (s | None, h | None, mean | None, rstd | None); gamma None: the sum only."""
    R, C = x.shape
    dev = x.device
    s = torch.empty(R, C, dtype=adt, device=dev) if want_s else None
    if gamma is None:
        _launch("gm3d_add_ln_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_fwd, _ptr(x), _ptr(y), _ptr(ybias), _ptr(rowscale),
                int(rps), _ptr(z), None, None, 0.0, _ptr(s), None, None, None, R, C, _DT[adt], _stream())
        return s, None, None, None
    h = h_out if h_out is not None else torch.empty(R, C, dtype=adt, device=dev)
    mean = torch.empty(R, dtype=torch.float32, device=dev)
    rstd = torch.empty(R, dtype=torch.float32, device=dev)
    _launch("gm3d_add_ln_fwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_fwd, _ptr(x), _ptr(y), _ptr(ybias), _ptr(rowscale),
            int(rps), _ptr(z), _ptr(gamma), _ptr(beta), float(eps), _ptr(s), _ptr(h), _ptr(mean), _ptr(rstd), R, C, _DT[adt], _stream())
    return s, h, mean, rstd


def _add_ln_bwd(dh, gin, x, mean, rstd, gamma, rowscale, rps, dy, partial, adt, acc=None, acc_mode=0):
    """-> dx (R,C) adt.  partial: a (rows, 3 C) slice; dy / acc written in place when given."""
    ref = dh if dh is not None else gin
    R, C = ref.shape
    dx = torch.empty(R, C, dtype=adt, device=ref.device)
    _launch("gm3d_add_ln_bwd", {"R": R, "C": C, "dtype": str(adt)}, lib.gm3d_add_ln_bwd_acc, _ptr(dh), _ptr(gin), _ptr(x), _ptr(mean),
            _ptr(rstd), _ptr(gamma), _ptr(rowscale), int(rps), _ptr(dx), _ptr(dy), _ptr(partial), 3, _ptr(acc), int(acc_mode), R, C,
            _DT[adt], _stream())
    return dx


def _attn_fwd(qkv, bits, B, T, H, hd, scale, out):
    if bits is None and hd == 64 and T <= 128:
        return fused._attention_fwd(qkv, B, T, H, scale, out=out)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _launch("gm3d_attention_masked_fwd", {"B": B, "T": T, "H": H, "HD": hd, "dtype": str(qkv.dtype)}, lib.gm3d_attention_masked_fwd,
            _ptr(qkv), _ptr(bits), _ptr(out), _ptr(lse), B, T, H, hd, float(scale), _DT[qkv.dtype], _stream())
    return out, lse


def _attn_bwd(qkv, a, da, lse, bits, B, T, H, hd, scale, dqkv):
    if bits is None and hd == 64 and T <= 128:
        return fused._attention_bwd(qkv, a, da, lse, B, T, H, scale, dqkv=dqkv)
    _launch("gm3d_attention_masked_bwd", {"B": B, "T": T, "H": H, "HD": hd, "dtype": str(qkv.dtype)}, lib.gm3d_attention_masked_bwd,
            _ptr(qkv), _ptr(bits), _ptr(a), _ptr(da), _ptr(lse), _ptr(dqkv), B, T, H, hd, float(scale), _DT[qkv.dtype], _stream())
    return dqkv


FUSE_GELU = True          # widths 192 / 384: bias + GELU (forward) and GELU' + bias-gradient partials (backward) in the GEMM epilogues (tests flip it)


def supported(x, blocks):
    C = x.shape[-1]
    H = blocks[0].attn.num_heads
    return (x.is_cuda and x.dim() == 3 and C % 8 == 0 and 8 <= C <= 512 and C % H == 0 and C // H in (16, 32, 64)
            and x.shape[1] <= 512 and x.dtype in (torch.float32, torch.bfloat16))


class MaskedStackFn(torch.autograd.Function):
    """args: x (B,T,C), pos (B,T,C), bits (B,T,ceil(T/32)) int32 | None, meta, then PER_BLOCK tensors per block (fused.block_params).
    meta: dict(num_heads, scale, eps, adt, dp=[(scale_attn | None, scale_mlp | None) per block], grad)."""

    @staticmethod
    def forward(ctx, x, pos, bits, meta, *params):
        with torch.autocast("cuda", enabled=False):
            return MaskedStackFn._forward(ctx, x, pos, bits, meta, *params)

    @staticmethod
    def _forward(ctx, x, pos, bits, meta, *params):
        B, T, C = x.shape
        nblk = len(params) // PER_BLOCK
        adt, H, scale, eps = meta["adt"], meta["num_heads"], meta["scale"], meta["eps"]
        hd = C // H
        R = B * T
        dev = x.device
        s = x.reshape(R, C).to(adt).contiguous()
        posa = pos.reshape(R, C).to(adt).contiguous()
        need = meta["grad"] and any(ctx.needs_input_grad)
        if need:      # operands of the weight-gradient products, stacked over the blocks
            H1 = torch.empty(nblk, R, C, dtype=adt, device=dev)
            A = torch.empty(nblk, R, C, dtype=adt, device=dev)
            H2 = torch.empty(nblk, R, C, dtype=adt, device=dev)
            GG = torch.empty(nblk, R, 4 * C, dtype=adt, device=dev)
        y = yb = rs = None
        saved = []
        for i in range(nblk):
            ln1w, ln1b, wqkv, wproj, bproj, ln2w, ln2b, w1, b1, w2, b2 = params[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            dp1, dp2 = meta["dp"][i]
            s1, h1, m1, r1 = _add_ln_fwd(s, y, yb, rs, T, posa, _c32(ln1w), _c32(ln1b), eps, adt, True, H1[i] if need else None)
            qkv = gemm.mm(h1, weight_cache.get(wqkv, adt))
            a, lse = _attn_fwd(qkv, bits, B, T, H, hd, scale, A[i] if need else torch.empty(R, C, dtype=adt, device=dev))
            p = gemm.mm(a, weight_cache.get(wproj, adt))
            s2, h2, m2, r2 = _add_ln_fwd(s1, p, _c32(bproj), dp1, T, None, _c32(ln2w), _c32(ln2b), eps, adt, True, H2[i] if need else None)
            W1 = weight_cache.get(w1, adt)
            if FUSE_GELU and adt == torch.bfloat16 and gemm.FUSE_GELU and gemm.dma_supported(h2, W1):
                # fc1 + bias + GELU in the product's epilogue (widths 192 / 384: N % 192 == 0, K % 64 == 0); the pre-activation is
                # written only when a backward follows
                f, g = gemm.linear_gelu_dma(h2, W1, _c32(b1), f_out=torch.empty(R, 4 * C, dtype=adt, device=dev) if need else None,
                                            g_out=GG[i] if need else None, bm=gemm.dma_bm(R))
            else:
                f = gemm.mm(h2, W1)
                g = fused.bias_gelu_fwd(f, _c32(b1), adt, g=GG[i] if need else None)
            y, yb, rs, s = gemm.mm(g, weight_cache.get(w2, adt)), _c32(b2), dp2, s2
            if need:
                saved += [s1, m1, r1, qkv, lse, s2, m2, r2, f]
        out, _, _, _ = _add_ln_fwd(s, y, yb, rs, T, None, None, None, 0.0, adt, True)
        if need:
            ctx.save_for_backward(H1, A, H2, GG, *params, *saved)
            ctx.bits = bits
        ctx.meta, ctx.shape, ctx.nblk, ctx.in_dtypes = meta, (B, T, C), nblk, (x.dtype, pos.dtype)
        return out.view(B, T, C)

    @staticmethod
    def backward(ctx, dout):
        with torch.autocast("cuda", enabled=False):
            return MaskedStackFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        meta, (B, T, C), nblk = ctx.meta, ctx.shape, ctx.nblk
        adt, H, scale = meta["adt"], meta["num_heads"], meta["scale"]
        hd, R = C // H, B * T
        bits = ctx.bits
        tens = ctx.saved_tensors
        H1, A, H2, GG = tens[:4]
        params = tens[4:4 + nblk * PER_BLOCK]
        saved = tens[4 + nblk * PER_BLOCK:]
        grads = [None] * (nblk * PER_BLOCK)
        dev = dout.device
        gin = dout.reshape(R, C).to(adt).contiguous()
        DO = torch.empty(nblk, R, C, dtype=adt, device=dev)          # output-side operands of the weight-gradient products
        DF = torch.empty(nblk, R, 4 * C, dtype=adt, device=dev)
        DP = torch.empty(nblk, R, C, dtype=adt, device=dev)
        DQ = torch.empty(nblk, R, 3 * C, dtype=adt, device=dev)
        # column-sum partials of every LayerNorm site (2i: LN1 of block i, 2i+1: LN2, 2 nblk: the tail) and every GELU site
        PLN = torch.empty(2 * nblk + 1, lib.gm3d_ln_plain_partial_rows(R), 3 * C, dtype=torch.float32, device=dev)
        SLN = torch.empty(2 * nblk + 1, 3 * C, dtype=torch.float32, device=dev)
        W2T, WPT, W1T, WQT = gemm.stacked_transposes(
            [[weight_cache.get(params[i * PER_BLOCK + k], adt) for i in range(nblk)] for k in (9, 3, 7, 2)])
        # fc2's input gradient x GELU'(f + b1) + fc1's bias-gradient partials in the product's epilogue where the tile shape allows
        dma_bwd = FUSE_GELU and adt == torch.bfloat16 and gemm.FUSE_GELU_BWD and gemm.dma_supported(gin, W2T[0])
        bm_bwd = gemm.dma_bm(R)
        PGL = torch.empty(nblk, (R + bm_bwd - 1) // bm_bwd if dma_bwd else lib.gm3d_gelu_partial_rows(R), 4 * C, dtype=torch.float32, device=dev)
        SGL = torch.empty(nblk, 4 * C, dtype=torch.float32, device=dev)
        G = _add_ln_bwd(None, gin, None, None, None, None, meta["dp"][nblk - 1][1], T, DO[nblk - 1], PLN[2 * nblk], adt)
        db2 = SLN[2 * nblk, 2 * C:]
        dpos = torch.empty(R, C, dtype=adt, device=dev)
        for i in range(nblk - 1, -1, -1):
            ln1w, ln1b, wqkv, wproj, bproj, ln2w, ln2b, w1, b1, w2, b2 = params[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            s1, m1, r1, qkv, lse, s2, m2, r2, f = saved[i * 9:(i + 1) * 9]
            gi = grads[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            gi[10] = db2
            if dma_bwd:
                gemm.linear_gelu_bwd_dma(DO[i], W2T[i], f, _c32(b1), DF[i], PGL[i], bm=bm_bwd)
            else:
                dg = gemm.mm(DO[i], W2T[i])
                fused.bias_gelu_bwd(dg, f, _c32(b1), adt, df=DF[i], partial=PGL[i])
            gi[8] = SGL[i]
            dh2 = gemm.mm(DF[i], W1T[i])
            G = _add_ln_bwd(dh2, G, s2, m2, r2, _c32(ln2w), meta["dp"][i][0], T, DP[i], PLN[2 * i + 1], adt)
            gi[5], gi[6], gi[4] = SLN[2 * i + 1, :C], SLN[2 * i + 1, C:2 * C], SLN[2 * i + 1, 2 * C:]
            da = gemm.mm(DP[i], WPT[i])
            _attn_bwd(qkv, A[i], da, lse, bits, B, T, H, hd, scale, DQ[i])
            dh1 = gemm.mm(DQ[i], WQT[i])
            G = _add_ln_bwd(dh1, G, s1, m1, r1, _c32(ln1w), meta["dp"][i - 1][1] if i > 0 else None, T, DO[i - 1] if i > 0 else None,
                            PLN[2 * i], adt, acc=dpos, acc_mode=1 if i == nblk - 1 else 2)
            gi[0], gi[1] = SLN[2 * i, :C], SLN[2 * i, C:2 * C]
            db2 = SLN[2 * i, 2 * C:]
            grads[i * PER_BLOCK:(i + 1) * PER_BLOCK] = gi
        fused.finish_batched(PLN, SLN)
        fused.finish_batched(PGL, SGL)
        # all weight gradients of the stack: ONE launch, into the parameters' slots of the flat gradient buffer where they are
        # adjacent there (optim._kind_key) and hold no gradient yet
        from .optim import grad_slots

        def slot(k):
            ps = [params[i * PER_BLOCK + k] for i in range(nblk)]
            return grad_slots.stacked(ps) if dev.type == "cuda" and all(p.grad is None for p in ps) else None
        gw2, gw1, gwp, gwq = fused._wgrad_many([(DO, GG, slot(9)), (DF, H2, slot(7)), (DP, A, slot(3)), (DQ, H1, slot(2))])
        for i in range(nblk):
            grads[i * PER_BLOCK + 9], grads[i * PER_BLOCK + 7] = gw2[i], gw1[i]
            grads[i * PER_BLOCK + 3], grads[i * PER_BLOCK + 2] = gwp[i], gwq[i]
        for j, (gr, p) in enumerate(zip(grads, params)):
            if gr is not None and gr.dtype != p.dtype:
                grads[j] = gr.to(p.dtype)
        return (G.view(B, T, C).to(ctx.in_dtypes[0]), dpos.view(B, T, C).to(ctx.in_dtypes[1]), None, None) + tuple(grads)


def run_stack(blocks, x, pos, bits, training, adt):
    """blocks: MaskedBlock modules (norm1, attn.qkv / proj, norm2, mlp.fc1 / fc2, drop_path); x, pos (B,T,C)."""
    from . import models_mae_learn_loss as M
    blocks = list(blocks)
    B = x.shape[0]
    probs = []
    for b in blocks:
        p = b.drop_path.drop_prob if isinstance(b.drop_path, M.DropPath) else 0.0
        probs += [p, p]
    scales = M.drop_path_scales(B, probs, training, x.device)
    meta = {"num_heads": blocks[0].attn.num_heads, "scale": blocks[0].attn.scale, "eps": blocks[0].norm1.eps, "adt": adt,
            "dp": [(scales[2 * i], scales[2 * i + 1]) for i in range(len(blocks))], "grad": torch.is_grad_enabled()}
    params = []
    for b in blocks:
        params += block_params(b)
    return MaskedStackFn.apply(x, pos, bits, meta, *params)


# ----------------------------------------------------------------------------- visible-first order
_overflow = {}


def overflow_flag(device):
    """(1,) int32 on `device`: set by gm3d_partition_visible when a cloud had more visible tokens than the caller's static bound
    (the compacted pass then dropped tokens).  The engine / tests read it; it is never cleared by the kernels."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _overflow:
        _overflow[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _overflow[key]


fused._streams.before_capture(overflow_flag)


def partition_visible(masked, Tc):
    """masked (B,T) bool / uint8 (True = masked) -> dict(perm_c, perm_v (B,Tc) int32, inv_v, inv_m (B,T) int32, vis_c (B,Tc) uint8);
    include/gm3d.h gm3d_partition_visible."""
    B, T = masked.shape
    m = masked.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    dev = masked.device
    i32 = dict(dtype=torch.int32, device=dev)
    out = {"perm_c": torch.empty(B, Tc, **i32), "perm_v": torch.empty(B, Tc, **i32), "inv_v": torch.empty(B, T, **i32),
           "inv_m": torch.empty(B, T, **i32), "vis_c": torch.empty(B, Tc, dtype=torch.uint8, device=dev), "T": T, "Tc": Tc}
    _launch("gm3d_partition_visible", {"B": B, "T": T}, lib.gm3d_partition_visible, _ptr(m), B, T, Tc, _ptr(out["perm_c"]),
            _ptr(out["perm_v"]), _ptr(out["inv_v"]), _ptr(out["inv_m"]), _ptr(out["vis_c"]), _ptr(overflow_flag(dev)), _stream())
    return out


def select_rows(a, idx, alt=None):
    """out (B,T,C): row t = a[b, idx[b,t]] where idx >= 0, else alt[b,t] (zeros when alt is None).  No autograd."""
    B, Ta, C = a.shape
    T = idx.shape[1]
    a = a.contiguous()
    if alt is not None:
        alt = alt.to(a.dtype).contiguous()
    out = torch.empty(B, T, C, dtype=a.dtype, device=a.device)
    _launch("gm3d_select_rows", {"B": B, "T": T, "C": C}, lib.gm3d_select_rows, _ptr(a), _ptr(idx), _ptr(alt), _ptr(out), B, Ta, T, C,
            a.element_size(), _stream())
    return out


class CompactFn(torch.autograd.Function):
    """x (B,T,C) -> (B,Tc,C): the rows perm_c names (visible tokens first).  Backward: every visible token takes its slot's gradient;
    filler slots are never read downstream, their gradient is dropped."""

    @staticmethod
    def forward(ctx, x, part):
        ctx.part = part
        return select_rows(x, part["perm_c"])

    @staticmethod
    def backward(ctx, dy):
        return select_rows(dy, ctx.part["inv_v"]), None


class MergeFn(torch.autograd.Function):
    """(y_c (B,Tc,C), tok (B,T,C)) -> (B,T,C): a visible token takes its encoded row from the compact order, a masked one keeps tok."""

    @staticmethod
    def forward(ctx, y_c, tok, part):
        ctx.part = part
        ctx.dts = (y_c.dtype, tok.dtype)
        return select_rows(y_c, part["inv_v"], tok.to(y_c.dtype))

    @staticmethod
    def backward(ctx, dout):
        part = ctx.part
        d_yc = select_rows(dout, part["perm_v"])
        d_tok = select_rows(dout, part["inv_m"])
        return d_yc.to(ctx.dts[0]), d_tok.to(ctx.dts[1]), None
