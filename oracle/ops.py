"""ctypes/torch front-end of the CPU oracle (oracle/gm3d_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gm3d_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED for FPS / KNN / Chamfer (third-party sources absent from the
reference tree, no reference tests): SURVEY.md section 8c.

The wrappers mirror the reference-side call signatures so the reference's own
model file can run on top of them (tests/golden/make_golden.py):
  furthest_point_sample / gather_operation -> pointnet2_ops.pointnet2_utils
      (call sites Point-MAE_SA3D/models_mae_learn_loss.py:931-932)
  KNN                                       -> knn_cuda.KNN (models_mae_learn_loss.py:924,946)
  ChamferDistanceL2 / L1                    -> extensions.chamfer_dist (models_mae_learn_loss.py:188,407)
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgm3d_oracle.so")


def build(force=False):
    """Compile the C oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "gm3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous().cpu()


# --------------------------------------------------------------------------- FPS / gather
def furthest_point_sample(xyz, npoint):
    """(B,N,3) f32 -> (B,npoint) int32."""
    x = _f32(xyz)
    B, N, _ = x.shape
    idx = torch.empty(B, npoint, dtype=torch.int32)
    lib().oracle_fps(_p(x), B, N, int(npoint), _p(idx))
    return idx


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        f = _f32(feat)
        i = idx.to(torch.int32).contiguous().cpu()
        B, C, N = f.shape
        M = i.shape[1]
        out = torch.empty(B, C, M, dtype=torch.float32)
        lib().oracle_gather(_p(f), _p(i), B, C, N, M, _p(out))
        ctx.save_for_backward(i)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, g):
        (i,) = ctx.saved_tensors
        g = _f32(g)
        B, C, M = g.shape
        gf = torch.empty(B, C, ctx.N, dtype=torch.float32)
        lib().oracle_gather_grad(_p(g), _p(i), B, C, ctx.N, M, _p(gf))
        return gf, None


def gather_operation(features, idx):
    """(B,C,N) f32, (B,M) int32 -> (B,C,M)."""
    return _Gather.apply(features, idx)


# --------------------------------------------------------------------------- KNN / group
def knn(ref, query, k):
    """ref (B,N,3), query (B,G,3) -> (dist (B,G,k) f32 [sqrt applied], idx (B,G,k) int64)."""
    r, q = _f32(ref), _f32(query)
    B, N, _ = r.shape
    G = q.shape[1]
    dist = torch.empty(B, G, k, dtype=torch.float32)
    idx = torch.empty(B, G, k, dtype=torch.int64)
    lib().oracle_knn(_p(r), _p(q), B, N, G, int(k), _p(dist), _p(idx))
    return dist, idx


class KNN:
    """knn_cuda.KNN(k, transpose_mode=True) stand-in; runs under no_grad like upstream."""

    def __init__(self, k, transpose_mode=False):
        assert transpose_mode, "the hot path only uses transpose_mode=True"
        self.k = k

    def __call__(self, ref, query):
        with torch.no_grad():
            return knn(ref, query, self.k)


def group(xyz, center, idx):
    x, c = _f32(xyz), _f32(center)
    i = idx.to(torch.int64).contiguous().cpu()
    B, N, _ = x.shape
    G, k = i.shape[1], i.shape[2]
    nb = torch.empty(B, G, k, 3, dtype=torch.float32)
    nbo = torch.empty(B, G, k, 3, dtype=torch.float32)
    lib().oracle_group(_p(x), _p(c), _p(i), B, N, G, k, _p(nb), _p(nbo))
    return nb, nbo


# --------------------------------------------------------------------------- Chamfer
class _Chamfer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        a, b = _f32(xyz1), _f32(xyz2)
        P, n, _ = a.shape
        m = b.shape[1]
        d1 = torch.empty(P, n); d2 = torch.empty(P, m)
        i1 = torch.empty(P, n, dtype=torch.int32); i2 = torch.empty(P, m, dtype=torch.int32)
        lib().oracle_chamfer_fwd(_p(a), _p(b), P, n, m, _p(d1), _p(d2), _p(i1), _p(i2))
        ctx.save_for_backward(a, b, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, g1, g2, _gi1, _gi2):
        a, b, i1, i2 = ctx.saved_tensors
        P, n, _ = a.shape
        m = b.shape[1]
        g1 = _f32(g1) if g1 is not None else torch.zeros(P, n)
        g2 = _f32(g2) if g2 is not None else torch.zeros(P, m)
        ga = torch.empty_like(a); gb = torch.empty_like(b)
        lib().oracle_chamfer_bwd(_p(a), _p(b), _p(i1), _p(i2), _p(g1), _p(g2), P, n, m, _p(ga), _p(gb))
        return ga, gb


def chamfer(xyz1, xyz2):
    """-> dist1 (P,n), dist2 (P,m), idx1, idx2."""
    return _Chamfer.apply(xyz1, xyz2)


class ChamferDistanceL2(torch.nn.Module):
    """reduction='per_point' is the GM3D-local variant assumed in SURVEY.md 0.3 / 8(a10):
    (P,n) tensor dist1+dist2 (needs n==m); 'mean' is upstream's scalar mean(d1)+mean(d2)."""

    def __init__(self, reduction="per_point"):
        super().__init__()
        self.reduction = reduction

    def forward(self, xyz1, xyz2):
        d1, d2, _, _ = chamfer(xyz1, xyz2)
        if self.reduction == "per_point":
            return d1 + d2
        return d1.mean() + d2.mean()


class ChamferDistanceL1(torch.nn.Module):
    def forward(self, xyz1, xyz2):
        d1, d2, _, _ = chamfer(xyz1, xyz2)
        return (torch.sqrt(d1).mean() + torch.sqrt(d2).mean()) / 2


# --------------------------------------------------------------------------- numpy cross-check
def fps_numpy(xyz, npoint):
    """Independent fp32 NumPy statement of the same FPS rule (slow; pins the C code in
    tests/test_oracle.py).  Follows datasets/ModelNetDataset.py:25-46 with start index 0,
    fp32, the |p|^2<=1e-3 skip and first-max ties."""
    xyz = np.asarray(xyz, dtype=np.float32)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), dtype=np.int32)
    for b in range(B):
        p = xyz[b]
        mag = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]
        ok = mag > np.float32(1e-3)
        temp = np.full(N, 1e10, dtype=np.float32)
        old = 0
        for j in range(1, npoint):
            d = p - p[old]
            d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            temp = np.where(ok, np.minimum(temp, d), temp)
            cand = np.where(ok, temp, np.float32(-2.0))
            best = cand.max()
            old = int(np.argmax(cand)) if best > -1.0 else 0
            out[b, j] = old
    return out
