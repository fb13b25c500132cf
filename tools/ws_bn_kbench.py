"""second_conv.0 on the weight-stationary kernel: plain product vs the BatchNorm epilogues (eval: apply + ReLU, train: statistics), and
the streaming kernels they replace, at the teacher's 262,144 rows.   python tools/ws_bn_kbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm
from gm3d_amd._capi import lib
from gm3d_amd.ops import _ptr, _stream

M, K, N = 262144, 256, 512
G = M // 32
g = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(M, K, device="cuda", generator=g).bfloat16() for _ in range(2)]
w = (torch.randn(N, K, device="cuda", generator=g) / 16).bfloat16()
t = torch.randn(G, N, device="cuda", generator=g).bfloat16()
sc, sh = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
y0 = gemm.linear_tn_ws(xs[0], w)
a2 = torch.empty_like(y0)
part = torch.empty(lib.gm3d_embed_partial_rows(1, G, N), 2 * N, device="cuda")


def timed(fn, n=20):
    for _ in range(3):
        fn(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("product                      %.1f us" % timed(lambda i: gemm.linear_tn_ws(xs[i & 1], w)))
print("product + BN apply + ReLU    %.1f us" % timed(lambda i: gemm.linear_ws_bn_apply(xs[i & 1], w, t, sc, sh)))
print("product + statistics         %.1f us" % timed(lambda i: gemm.linear_ws_bn_stats(xs[i & 1], w, t)))
print("bn_bcast_apply_relu alone    %.1f us" % timed(lambda i: lib.gm3d_bn_bcast_apply_relu(_ptr(y0), _ptr(t), _ptr(sc), _ptr(sh), _ptr(a2), G, 32, N, 0.0, 1, _stream())))
print("bn_bcast_stats alone         %.1f us" % timed(lambda i: lib.gm3d_bn_bcast_stats(_ptr(y0), _ptr(t), G, 32, N, _ptr(part), 1, _stream())))
