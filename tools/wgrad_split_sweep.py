"""NT weight-gradient kernel: time per shape as a function of the row split (incl. the sum of the partial slabs).  python tools/wgrad_split_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm
from gm3d_amd._capi import lib
from tools.wgrad_bench import timeit, shapes

for nb, R, N, K in shapes:
    dy = torch.randn(nb, R, N, device="cuda").bfloat16()
    x = torch.randn(nb, R, K, device="cuda").bfloat16()
    out = torch.empty(nb, N, K, device="cuda")
    res = []
    for s in (1, 2, 4, 8, 16, 32, 64):
        if R % (32 * s) or R // s < 256:
            continue
        res.append((s, timeit(lambda: gemm.wgrad_nt(dy, x, out, splits=s))))
    auto = lib.gm3d_gemm_nt_splits(nb, R, N, K)
    print("%-26s tiles %4d  auto %2d   %s" % (str((nb, R, N, K)), nb * (N // 128) * (K // 128), auto, "  ".join("s=%d: %.1f" % r for r in res)))
