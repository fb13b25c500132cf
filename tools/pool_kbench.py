"""conv + max-pool epilogue of the mini-PointNet on the kernels that carry it (csrc/gemm.hip, gemm_dma.hip, gemm_ws.hip): captured
trains of launches, HIP events.   python tools/pool_kbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm
from tools.gemm_kbench import train

dev = "cuda"
for groups, K, N, after, rows in ((8192, 128, 256, False, True), (8192, 512, 384, True, False), (3200, 512, 384, True, False)):
    M = groups * 32
    sets = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(), torch.randn(N, device=dev))
            for _ in range(2)]
    res = {}
    for name, ws, dma in (("own", False, False), ("dma", False, True), ("ws", True, True)):
        gemm.USE_WS, gemm.POOL_ON_DMA = ws, dma
        res[name] = train(lambda x, w, b: gemm.linear_pool(x, w, b, after, rows), sets, iters=20)
    gemm.USE_WS = gemm.POOL_ON_DMA = True
    hbm = 2.0 * (M * K + (M * N if rows else 0) + groups * N) / 1e6
    print("groups=%5d K=%3d N=%3d rows=%d | " % (groups, K, N, rows) + "  ".join("%s %.1f us" % kv for kv in res.items())
          + " | HBM bytes %.0f MB -> %.1f us at 6 TB/s" % (hbm, hbm / 6.0))
