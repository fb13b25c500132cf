"""Do two independent kernel chains captured as parallel branches of one hipGraph overlap?  fc1(+GELU) -> fc2 chains of 12 blocks:
  one chain of 2M rows  |  one chain of M rows  |  two chains of M rows on two streams (fork / join inside the capture)
    python tools/chain_overlap.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blocks = 12
dev = torch.device("cuda")
g0 = torch.Generator(device="cuda").manual_seed(0)
xs = [(torch.randn(2 * M, 384, device=dev, generator=g0) * 0.5).bfloat16() for _ in range(2)]
W1 = [(torch.randn(1536, 384, device=dev, generator=g0) * 0.05).bfloat16() for _ in range(blocks)]
W2 = [(torch.randn(384, 1536, device=dev, generator=g0) * 0.02).bfloat16() for _ in range(blocks)]
b1 = torch.zeros(1536, device=dev)
b2 = torch.zeros(384, device=dev)
side = torch.cuda.Stream()


def chain(x):
    h = x
    for i in range(blocks):
        _, f = gemm.linear_gelu_dma(h, W1[i], b1, bm=gemm.dma_bm(h.shape[0]))
        h = gemm.mm(f, W2[i], b2)
    return h


def one_big():
    return chain(xs[0])


def one_half():
    return chain(xs[0][:M])


def two_halves():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        a = chain(xs[1][:M])
    b = chain(xs[0][:M])
    main.wait_stream(side)
    return a, b


def timed(fn, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, fn in (("one chain of %d rows" % (2 * M), one_big), ("one chain of %d rows" % M, one_half), ("two chains of %d rows, two streams" % M, two_halves)):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            keep = fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    print("%-40s %.1f us per replay (24 GEMMs per chain)" % (name, timed(g.replay)), flush=True)
# eager, two streams (the host keeps both queues fed?)
for _ in range(3):
    two_halves()
print("%-40s %.1f us eager" % ("two chains, two streams", timed(two_halves)))
print("%-40s %.1f us eager" % ("one chain of %d rows" % M, timed(one_half)))
