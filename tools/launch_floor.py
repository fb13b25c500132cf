"""What does one more DEPENDENT launch cost inside a replayed hipGraph on this stack?  Chains of N tiny kernels (one workgroup writing
one float), of 5 us-class kernels (the 3200 x 384 x 384 ring GEMM) and both mixed, as one graph on one stream.   python tools/launch_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm

dev = "cuda"


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def graph_of(body):
    s = torch.cuda.Stream()
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            body()
    return g


tiny = torch.zeros(64, device=dev)
x = torch.randn(3200, 384, device=dev).bfloat16()
w = (torch.randn(384, 384, device=dev) * 0.05).bfloat16()
outs = [torch.empty(3200, 384, device=dev, dtype=torch.bfloat16) for _ in range(4)]
for n in (50, 200, 800):
    g = graph_of(lambda: [tiny.add_(1.0) for _ in range(n)])
    t = timeit(lambda: g.replay())
    print("%4d dependent tiny launches (64-element add_): %8.1f us  = %.2f us per launch" % (n, t, t / n))
for n in (50, 200):
    g = graph_of(lambda: [gemm.linear_tn_ring(x, w, out=outs[i % 4], bm=64) for i in range(n)])
    t = timeit(lambda: g.replay())
    print("%4d ring GEMMs 3200x384x384 (150 workgroups):  %8.1f us  = %.2f us per launch" % (n, t, t / n))
