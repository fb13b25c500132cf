"""KNN grouping kernel timing (captured train of launches, HIP events): python tools/knn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import ops

for B, N, G, k in [(128, 1024, 64, 32), (32, 1024, 64, 32), (128, 2048, 512, 16), (16, 8192, 64, 32)]:
    xyz = torch.randn(B, N, 3, device="cuda")
    ctr = xyz[:, :G].contiguous()
    fn = lambda: ops.knn_group(xyz, ctr, k)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("B=%d N=%d G=%d k=%d  %.1f us" % (B, N, G, k, e0.elapsed_time(e1) * 1e3 / 20))
