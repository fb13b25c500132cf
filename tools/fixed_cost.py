"""What is in the fixed ~5 us of a dependent GEMM launch?  Chains of 24 ring-kernel products (x -> x W^T -> ...), one graph each:
tiny M (one or a few workgroups: no bandwidth, no tail) for K = 384 and K = 1536, and the step's 3328 / 8192 rows for reference;
plus a chain of LayerNorm-like streaming kernels at the same sizes.   python tools/fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm

dev = torch.device("cuda")
g0 = torch.Generator(device="cuda").manual_seed(0)


def timed(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def graph_of(body):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            keep = body()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    return g, keep


for K in (384, 1536):
    Ws = [(torch.randn(384, K, device=dev, generator=g0) * 0.05).bfloat16() for _ in range(4)]
    Wb = [(torch.randn(K, 384, device=dev, generator=g0) * 0.05).bfloat16() for _ in range(4)]
    for M in (64, 512, 3328, 8192):
        x = (torch.randn(M, K, device=dev, generator=g0) * 0.5).bfloat16()

        def body():
            h = x
            for i in range(12):
                h = gemm.linear_tn_ring(h, Ws[i % 4], bm=64 if M <= 4096 else 128)      # (M,K) -> (M,384)
                h = gemm.linear_tn_ring(h, Wb[i % 4], bm=64 if M <= 4096 else 128) if K != 384 else gemm.linear_tn_ring(h, Ws[(i + 1) % 4], bm=64 if M <= 4096 else 128)
            return h
        g, keep = graph_of(body)
        print("ring GEMM chain  K=%4d (and back)  M=%5d   %.2f us per launch" % (K, M, timed(g.replay) / 24), flush=True)
tiny = torch.zeros(64, device=dev)
g, _ = graph_of(lambda: [tiny.add_(1.0) for _ in range(24)])
print("24 dependent 64-element add_ launches: %.2f us per launch" % (timed(g.replay) / 24))
