"""How much do two independent chains of under-filled kernels overlap on this stack?  Chains of ring GEMMs (4096 x 1536 -> 384: 192
workgroups for 256 CUs, ~11.5 us each), 24 launches per chain:
  serial      both chains on one stream, one graph
  branches    one graph, the chains captured on two forked streams (what the step does for the teacher's halves / the decoders)
  two graphs  one graph per chain, replayed on two streams
    python tools/graph_branch_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm

dev = "cuda"
M, K, N, L = 4096, 1536, 384, 24


def mk():
    return (torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(),
            torch.empty(M, N, device=dev, dtype=torch.bfloat16))


A, B_ = [mk() for _ in range(4)], [mk() for _ in range(4)]


def chain(sets):
    for i in range(L):
        x, w, o = sets[i % 4]
        gemm.linear_tn_ring(x, w, out=o, bm=64)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


chain(A); chain(B_); torch.cuda.synchronize()
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
# serial
g_serial = torch.cuda.CUDAGraph()
with torch.cuda.stream(s0):
    with torch.cuda.graph(g_serial, stream=s0):
        chain(A); chain(B_)
# branches
g_br = torch.cuda.CUDAGraph()
with torch.cuda.stream(s0):
    with torch.cuda.graph(g_br, stream=s0):
        s1.wait_stream(s0)
        with torch.cuda.stream(s1):
            chain(B_)
        chain(A)
        s0.wait_stream(s1)
# branches, launches interleaved in capture order
g_il = torch.cuda.CUDAGraph()
with torch.cuda.stream(s0):
    with torch.cuda.graph(g_il, stream=s0):
        s1.wait_stream(s0)
        for i in range(L):
            x, w, o = A[i % 4]
            gemm.linear_tn_ring(x, w, out=o, bm=64)
            with torch.cuda.stream(s1):
                x, w, o = B_[i % 4]
                gemm.linear_tn_ring(x, w, out=o, bm=64)
        s0.wait_stream(s1)
# two graphs
gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.stream(s1):
    with torch.cuda.graph(gA, stream=s1):
        chain(A)
with torch.cuda.stream(s2):
    with torch.cuda.graph(gB, stream=s2):
        chain(B_)
torch.cuda.synchronize()


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        gA.replay()
    with torch.cuda.stream(s2):
        gB.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)


one = timeit(lambda: gA.replay())
print("one chain alone (24 launches)          %7.1f us  (%.1f us per launch)" % (one, one / L))
for name, fn in (("serial, one graph (48 launches)", lambda: g_serial.replay()), ("two branches of one graph", lambda: g_br.replay()),
                 ("two branches, interleaved capture", lambda: g_il.replay()), ("two graphs on two streams", two)):
    t = timeit(fn)
    print("%-38s %7.1f us  = %.2f x one chain" % (name, t, t / one))
