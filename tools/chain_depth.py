"""Ring depth of csrc/gemm_ring.hip in a DEPENDENT chain (operands just written by the previous kernel, i.e. not L2-hot as in
tools/gemm_kbench.py): fc1(+GELU, LDS-DMA kernel) -> fc2 (ring kernel) x 12, replayed as one graph, for ring depths 2 / 3 / 4.
    python tools/chain_depth.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm

M = int(sys.argv[1]) if len(sys.argv) > 1 else 3328
blocks = 12
dev = torch.device("cuda")
g0 = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(M, 384, device=dev, generator=g0) * 0.5).bfloat16()
W1 = [(torch.randn(1536, 384, device=dev, generator=g0) * 0.05).bfloat16() for _ in range(blocks)]
W2 = [(torch.randn(384, 1536, device=dev, generator=g0) * 0.02).bfloat16() for _ in range(blocks)]
Wp = [(torch.randn(384, 384, device=dev, generator=g0) * 0.05).bfloat16() for _ in range(blocks)]
b1 = torch.zeros(1536, device=dev)
b2 = torch.zeros(384, device=dev)


def chain(bm):
    h = x
    for i in range(blocks):
        _, f = gemm.linear_gelu_dma(h, W1[i], b1, bm=gemm.dma_bm(M))
        h = gemm.linear_tn_ring(f, W2[i], b2, bm=bm)
        h = gemm.linear_tn_ring(h, Wp[i], b2, bm=bm)       # a K = 384 product as well (proj)
    return h


def timed(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for bm in (64, 128):
    for depth in (2, 3, 4):
        gemm.lib.gm3d_gemm_ring_set_depth(bm, depth)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            chain(bm)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                keep = chain(bm)
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(3):
            g.replay()
        print("M=%d  ring bm=%d depth=%d   %.1f us per block (fc1 + fc2 + proj)" % (M, bm, depth, timed(g.replay) / blocks), flush=True)
    gemm.lib.gm3d_gemm_ring_set_depth(bm, 2)
