"""Torch-facing operators over the C ABI (include/gm3d.h).

Names, argument meaning and error behaviour mirror the reference-side interfaces the
GM3D pretrain path imports:
  furthest_point_sample / gather_operation <- pointnet2_ops.pointnet2_utils
      (Point-MAE_SA3D/models_mae_learn_loss.py:931-932, utils/miscc.py:18-19)
  KNN                                       <- knn_cuda.KNN (models_mae_learn_loss.py:924,946)
  ChamferDistanceL2 / ChamferDistanceL1     <- extensions.chamfer_dist (models_mae_learn_loss.py:188,407)
plus two fused entry points the reference has no counterpart for (knn_group, attention).
PyTorch is plumbing only here: it owns device memory and the current HIP stream.
"""
import ctypes

import torch

from . import _capi
from ._capi import check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  bench.py installs one over its timed region for the roofline figures;
    when none is installed the wrappers add nothing.  `only` restricts timing to some kernel names."""

    def __init__(self, only=None):
        self.only = set(only) if only else None
        self.records = {}   # name -> [(start_event, end_event, meta)]

    def wants(self, name):
        return self.only is None or name in self.only

    @staticmethod
    def bracket_overhead_ms(n=200):
        """Median elapsed time of an EMPTY event bracket on the current stream: what start/stop recording itself adds to
        every measured launch (subtract it before comparing with rocprof's kernel durations)."""
        ev = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev)
        return ms[len(ms) // 2]

    def summary(self):
        """name -> {"launches", "avg_ms", "total_ms", "meta"} (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms), "meta": recs[0][2],
                         "per_launch": [(m, r[2]) for m, r in zip(ms, recs)]}
        return out


_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


def _launch(name, meta, fn, *args):
    """Call one C-ABI entry point, check its status, optionally bracket it with HIP events."""
    t = _timer
    if t is None or not t.wants(name):
        check(fn(*args), name)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    check(fn(*args), name)
    e.record()
    t.records.setdefault(name, []).append((s, e, meta))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require(t, dtype, name, contiguous=True):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a GPU tensor (gm3d_amd has no CPU fallback)" % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if contiguous and not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


# ------------------------------------------------------------------ FPS / gather
def fps(xyz, npoint, return_centers=True):
    """xyz (B,N,3) f32 contiguous -> (idx (B,npoint) int32, centers (B,npoint,3) f32 | None)."""
    _require(xyz, torch.float32, "xyz")
    if xyz.dim() != 3 or xyz.size(2) != 3:
        raise RuntimeError("xyz must be (B,N,3)")
    B, N, _ = xyz.shape
    idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    cen = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device) if return_centers else None
    _launch("gm3d_fps", {"B": B, "N": N, "npoint": int(npoint)}, lib.gm3d_fps, _ptr(xyz), B, N, int(npoint), _ptr(idx), _ptr(cen), _stream())
    return idx, cen


def furthest_point_sample(xyz, npoint):
    """pointnet2_utils.furthest_point_sample: (B,N,3) f32 -> (B,npoint) int32, non-differentiable."""
    with torch.no_grad():
        return fps(xyz.detach(), npoint, return_centers=False)[0]


class _GatherOperation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        _require(features, torch.float32, "features")
        _require(idx, torch.int32, "idx")
        B, C, N = features.shape
        M = idx.size(1)
        out = torch.empty(B, C, M, dtype=torch.float32, device=features.device)
        _launch("gm3d_gather_points", {"B": B, "C": C, "N": N, "M": M}, lib.gm3d_gather_points, _ptr(features), _ptr(idx), B, C, N, M, _ptr(out), _stream())
        ctx.save_for_backward(idx)
        ctx.N = N
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        g = grad_out.contiguous().float()
        B, C, M = g.shape
        gf = torch.empty(B, C, ctx.N, dtype=torch.float32, device=g.device)
        _launch("gm3d_gather_points_grad", {"B": B, "C": C, "N": ctx.N, "M": M}, lib.gm3d_gather_points_grad, _ptr(g), _ptr(idx), B, C, ctx.N, M, _ptr(gf), _stream())
        return gf, None


def gather_operation(features, idx):
    """pointnet2_utils.gather_operation: features (B,C,N) f32, idx (B,M) int32 -> (B,C,M)."""
    return _GatherOperation.apply(features, idx)


# ------------------------------------------------------------------ KNN / grouping
def knn(ref, query, k, return_dist=True):
    ref = ref.detach().contiguous()
    query = query.detach().contiguous()
    _require(ref, torch.float32, "ref")
    _require(query, torch.float32, "query")
    if ref.dim() != 3 or query.dim() != 3 or ref.size(2) != 3 or query.size(2) != 3 or ref.size(0) != query.size(0):
        raise RuntimeError("ref must be (B,N,3) and query (B,G,3)")
    B, N, _ = ref.shape
    G = query.size(1)
    idx = torch.empty(B, G, k, dtype=torch.int64, device=ref.device)
    dist = torch.empty(B, G, k, dtype=torch.float32, device=ref.device) if return_dist else None
    _launch("gm3d_knn", {"B": B, "N": N, "G": G, "k": int(k)}, lib.gm3d_knn, _ptr(ref), _ptr(query), B, N, G, int(k), _ptr(dist), _ptr(idx), _stream())
    return dist, idx


class KNN(torch.nn.Module):
    """knn_cuda.KNN(k, transpose_mode=True): (ref (B,N,3), query (B,G,3)) -> (dist, idx int64),
    ascending, computed under no_grad like upstream."""

    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        with torch.no_grad():
            if not self._t:  # upstream's default layout is (B,dim,N)
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            dist, idx = knn(ref, query, self.k)
            if not self._t:
                dist, idx = dist.transpose(1, 2).contiguous(), idx.transpose(1, 2).contiguous()
            return dist, idx


def knn_group(xyz, center, k, return_idx=True, return_org=True):
    """Fused Group.forward tail (models_mae_learn_loss.py:946-957):
    -> (neighborhood (B,G,k,3) centred, neighborhood_org | None, idx (B,G,k) int64 | None)."""
    xyz = xyz.detach()
    center = center.detach()
    _require(xyz, torch.float32, "xyz")
    _require(center, torch.float32, "center")
    B, N, _ = xyz.shape
    G = center.size(1)
    dev = xyz.device
    nb = torch.empty(B, G, k, 3, dtype=torch.float32, device=dev)
    nbo = torch.empty(B, G, k, 3, dtype=torch.float32, device=dev) if return_org else None
    idx = torch.empty(B, G, k, dtype=torch.int64, device=dev) if return_idx else None
    _launch("gm3d_knn_group", {"B": B, "N": N, "G": G, "k": int(k)}, lib.gm3d_knn_group, _ptr(xyz), _ptr(center), B, N, G, int(k), _ptr(idx), _ptr(nb), _ptr(nbo), _stream())
    return nb, nbo, idx


# ------------------------------------------------------------------ Chamfer
class _Chamfer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        a = xyz1.contiguous()
        b = xyz2.contiguous()
        _require(a, torch.float32, "xyz1")
        _require(b, torch.float32, "xyz2")
        if a.dim() != 3 or b.dim() != 3 or a.size(2) != 3 or b.size(2) != 3 or a.size(0) != b.size(0):
            raise RuntimeError("xyz1 must be (P,n,3) and xyz2 (P,m,3)")
        P, n, _ = a.shape
        m = b.size(1)
        dev = a.device
        d1 = torch.empty(P, n, dtype=torch.float32, device=dev)
        d2 = torch.empty(P, m, dtype=torch.float32, device=dev)
        i1 = torch.empty(P, n, dtype=torch.int32, device=dev)
        i2 = torch.empty(P, m, dtype=torch.int32, device=dev)
        _launch("gm3d_chamfer_fwd", {"P": P, "n": n, "m": m}, lib.gm3d_chamfer_fwd, _ptr(a), _ptr(b), P, n, m, _ptr(d1), _ptr(d2), _ptr(i1), _ptr(i2), _stream())
        ctx.save_for_backward(a, b, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, g1, g2, _gi1, _gi2):
        a, b, i1, i2 = ctx.saved_tensors
        P, n, _ = a.shape
        m = b.size(1)
        g1 = g1.contiguous().float() if g1 is not None else None
        g2 = g2.contiguous().float() if g2 is not None else None
        ga = torch.empty_like(a)
        gb = torch.empty_like(b)
        _launch("gm3d_chamfer_bwd", {"P": P, "n": n, "m": m}, lib.gm3d_chamfer_bwd, _ptr(a), _ptr(b), _ptr(i1), _ptr(i2), _ptr(g1), _ptr(g2), P, n, m,
                                   _ptr(ga), _ptr(gb), _stream())
        return ga, gb


def chamfer(xyz1, xyz2):
    """-> (dist1 (P,n), dist2 (P,m), idx1 int32, idx2 int32); differentiable wrt both clouds."""
    return _Chamfer.apply(xyz1, xyz2)


class ChamferDistanceL2(torch.nn.Module):
    """extensions.chamfer_dist.ChamferDistanceL2.

    reduction='per_point' (default) is the GM3D-local form the hot path needs
    (models_mae_learn_loss.py:407-412 reshapes the result to (N,-1,n)): a (P,n) tensor
    dist1 + dist2, n == m.  The exact combination is not recoverable from the reference
    tree -- parity unpinned.  reduction='mean' is the upstream scalar
    mean(dist1) + mean(dist2) used by models/Point_MAE.py:426.
    """

    def __init__(self, ignore_zeros=False, reduction="per_point"):
        super().__init__()
        self.ignore_zeros = ignore_zeros
        self.reduction = reduction

    def forward(self, xyz1, xyz2):
        if self.ignore_zeros and xyz1.size(0) == 1:  # upstream's batch-1 zero filter
            xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(0)
            xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(0)
        d1, d2, _, _ = chamfer(xyz1, xyz2)
        if self.reduction == "per_point":
            if d1.shape != d2.shape:
                raise RuntimeError("per_point Chamfer needs n == m")
            return d1 + d2
        return torch.mean(d1) + torch.mean(d2)


class ChamferDistanceL1(torch.nn.Module):
    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        if self.ignore_zeros and xyz1.size(0) == 1:
            xyz1 = xyz1[torch.sum(xyz1, dim=2).ne(0)].unsqueeze(0)
            xyz2 = xyz2[torch.sum(xyz2, dim=2).ne(0)].unsqueeze(0)
        d1, d2, _, _ = chamfer(xyz1, xyz2)
        return (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2


# ------------------------------------------------------------------ attention
_DT = {torch.float32: _capi.GM3D_F32, torch.bfloat16: _capi.GM3D_BF16}


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, num_heads, scale):
        qkv = qkv.contiguous()
        if qkv.dtype not in _DT:
            raise RuntimeError("attention supports float32 and bfloat16, got %s" % qkv.dtype)
        _require(qkv, None, "qkv")
        B, T, C3 = qkv.shape
        if C3 != 3 * num_heads * 64:
            raise RuntimeError("qkv last dim must be 3*num_heads*64")
        out = torch.empty(B, T, num_heads * 64, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=qkv.device)
        _launch("gm3d_attention_fwd", {"B": B, "T": T, "H": num_heads, "dtype": str(qkv.dtype)}, lib.gm3d_attention_fwd, _ptr(qkv), _ptr(out), _ptr(lse), B, T, num_heads, float(scale),
                                     _DT[qkv.dtype], _stream())
        ctx.save_for_backward(qkv, out, lse)
        ctx.num_heads, ctx.scale = num_heads, float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        dout = dout.contiguous().to(qkv.dtype)
        B, T, _ = qkv.shape
        dqkv = torch.empty_like(qkv)
        _launch("gm3d_attention_bwd", {"B": B, "T": T, "H": ctx.num_heads, "dtype": str(qkv.dtype)}, lib.gm3d_attention_bwd, _ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, T, ctx.num_heads,
                                     ctx.scale, _DT[qkv.dtype], _stream())
        return dqkv, None, None


def attention(qkv, num_heads, scale):
    """qkv (B,T,3*H*64) = output of the qkv Linear (timm layout (B,T,3,H,64)) -> (B,T,H*64).  T <= 128: one workgroup per (cloud,
    head) with the whole head in LDS; 128 < T <= 512 (cfgs/config_3.yaml: 256 groups): the flash-style kernel of the hierarchical
    encoder without a mask."""
    if qkv.shape[1] > 128:
        return _AttentionMasked.apply(qkv, None, num_heads, scale)
    return _Attention.apply(qkv, num_heads, scale)


def pack_mask(mask):
    """(B,T,T) bool, True = pair not allowed -> (B,T,ceil(T/32)) int32 bitset for attention_masked (bit j&31 of word j>>5)."""
    B, T, _ = mask.shape
    W = (T + 31) // 32
    m = torch.ones(B, T, W * 32, dtype=torch.int64, device=mask.device)      # keys past T: blocked (the kernels force them anyway)
    m[:, :, :T] = mask.to(torch.int64)
    weights = (torch.ones(32, dtype=torch.int64, device=mask.device) << torch.arange(32, device=mask.device))
    words = (m.view(B, T, W, 32) * weights).sum(-1)                     # 0 .. 2^32-1
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()


def radius_mask_bits(center, vis, radius, masked=None):
    """center (B,G,3) f32, vis (B,G) bool | None -> (B,G,ceil(G/32)) int32 bitset: pair (i,j) blocked iff i or j invisible or their
    centres are >= radius apart (== pack_mask(~(vis_i & vis_j) | dist2 >= radius^2), in one launch and without the (B,G,G) tensors).
    masked (B,G) bool (True = NOT visible) may be given instead of vis (no bitwise_not, no converting copy)."""
    center = center.detach().contiguous()
    _require(center, torch.float32, "center")
    B, G, _ = center.shape
    inv = 0
    if masked is not None and vis is None:
        v = masked.contiguous()
        v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
        inv = 1
    elif vis is not None:
        v = vis.contiguous()
        v = v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)
    else:
        v = None
    bits = torch.empty(B, G, (G + 31) // 32, dtype=torch.int32, device=center.device)
    _launch("gm3d_radius_mask_bits", {"B": B, "G": G}, lib.gm3d_radius_mask_bits_m, _ptr(center), _ptr(v), inv, float(radius), B, G,
            _ptr(bits), _stream())
    return bits


class _AttentionMasked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bits, num_heads, scale):
        qkv = qkv.contiguous()
        if qkv.dtype not in _DT:
            raise RuntimeError("attention supports float32 and bfloat16, got %s" % qkv.dtype)
        _require(qkv, None, "qkv")
        B, T, C3 = qkv.shape
        hd = C3 // (3 * num_heads)
        if C3 != 3 * num_heads * hd or hd not in (16, 32, 64):
            raise RuntimeError("qkv last dim must be 3*num_heads*head_dim with head_dim 16, 32 or 64")
        if bits is not None and (bits.dtype != torch.int32 or tuple(bits.shape) != (B, T, (T + 31) // 32) or not bits.is_contiguous()
                                 or not bits.is_cuda):
            raise RuntimeError("mask bitset must be a contiguous int32 (B,T,ceil(T/32)) GPU tensor (ops.pack_mask)")
        out = torch.empty(B, T, num_heads * hd, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=qkv.device)
        _launch("gm3d_attention_masked_fwd", {"B": B, "T": T, "H": num_heads, "HD": hd, "dtype": str(qkv.dtype)},
                lib.gm3d_attention_masked_fwd, _ptr(qkv), _ptr(bits), _ptr(out), _ptr(lse), B, T, num_heads, hd, float(scale),
                _DT[qkv.dtype], _stream())
        ctx.save_for_backward(qkv, out, lse, bits)
        ctx.num_heads, ctx.scale, ctx.hd = num_heads, float(scale), hd
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, bits = ctx.saved_tensors
        dout = dout.contiguous().to(qkv.dtype)
        B, T, _ = qkv.shape
        dqkv = torch.empty_like(qkv)
        _launch("gm3d_attention_masked_bwd", {"B": B, "T": T, "H": ctx.num_heads, "HD": ctx.hd, "dtype": str(qkv.dtype)},
                lib.gm3d_attention_masked_bwd, _ptr(qkv), _ptr(bits), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), B, T,
                ctx.num_heads, ctx.hd, ctx.scale, _DT[qkv.dtype], _stream())
        return dqkv, None, None, None


def attention_masked(qkv, bits, num_heads, scale):
    """qkv (B,T,3*H*hd), hd in {16,32,64}, T <= 512; bits = pack_mask(mask) of a SYMMETRIC (B,T,T) bool mask (True = not allowed)
    or None -> (B,T,H*hd).  A query with no allowed key yields zeros.  (The hierarchical Point-M2AE encoder's attention.)"""
    return _AttentionMasked.apply(qkv, bits, num_heads, scale)
