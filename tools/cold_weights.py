"""How much of a dependent GEMM chain's time is cold operands?  A chain fc1(+GELU) -> fc2 -> fc1 -> ... of `blocks` blocks on M rows
(384 -> 1536 -> 384, bf16), captured as a hipGraph, replayed:
  hot    every block uses the SAME two weight matrices (they stay in the L2s)
  warm   every block has its own weights (12 x 2.4 MB: out of the 4 MB L2s, inside the 256 MB memory-side cache)
  cold   as warm, with a 2 GB fill between the replays (weights come from HBM); the fill is timed alone and subtracted
    python tools/cold_weights.py [M]"""
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
b1 = torch.zeros(1536, device=dev)
b2 = torch.zeros(384, device=dev)
big = torch.empty(1 << 30, dtype=torch.int16, device=dev)


def chain(same):
    h = x
    for i in range(blocks):
        j = 0 if same else i
        _, f = gemm.linear_gelu_dma(h, W1[j], b1, bm=gemm.dma_bm(M))
        h = gemm.mm(f, W2[j], b2)
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


res = {}
for name, same in (("hot", True), ("warm", False)):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(same)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(same)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    res[name] = timed(g.replay)
    if not same:
        fill = timed(lambda: big.fill_(1))
        both = timed(lambda: (big.fill_(1), g.replay()))
        res["cold"] = both - fill
        res["fill"] = fill
n = 2 * blocks
print("M=%d, %d dependent GEMMs: hot %.1f us (%.2f per GEMM)   warm %.1f (%.2f)   cold %.1f (%.2f)   [2 GB fill alone %.1f us]"
      % (M, n, res["hot"], res["hot"] / n, res["warm"], res["warm"] / n, res["cold"], res["cold"] / n, res["fill"]))
