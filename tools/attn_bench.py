"""Micro-benchmark of the attention kernels at the step's shapes (captured train of launches replayed, HIP events).
    python tools/attn_bench.py            GM3D_HIP_LIB=<other .so> selects another build for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd._capi import lib, check
from gm3d_amd.ops import _ptr, _stream

dev = torch.device("cuda")
NSET = 4


def bench(name, call, sets, iters=40):
    for s in sets:
        call(*s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(iters):
                call(*sets[i % len(sets)])
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)


if __name__ == "__main__":
    for (B, T, H) in ((64, 64, 6), (128, 64, 6), (128, 25, 6), (32, 65, 6), (128, 128, 6)):
        for dt, code in ((torch.bfloat16, 1),):
            def mk():
                qkv = (torch.randn(B, T, 3 * H * 64, device=dev) * 0.5).to(dt)
                return (qkv, torch.empty(B * T, H * 64, device=dev, dtype=dt), torch.empty(B, H, T, device=dev),
                        (torch.randn(B * T, H * 64, device=dev) * 0.1).to(dt), torch.empty_like(qkv))
            sets = [mk() for _ in range(NSET)]
            f = bench("fwd", lambda q, o, l, do, dq: check(lib.gm3d_attention_fwd(_ptr(q), _ptr(o), _ptr(l), B, T, H, 0.125, code, _stream()), "f"), sets)
            b = bench("bwd", lambda q, o, l, do, dq: check(lib.gm3d_attention_bwd(_ptr(q), _ptr(o), _ptr(do), _ptr(l), _ptr(dq), B, T, H, 0.125, code,
                                                                                  _stream()), "b"), sets)
            fl = B * H * 4.0 * T * T * 64
            print("B=%3d T=%3d  fwd %6.2f us (%5.1f TFLOP/s)   bwd %6.2f us (%5.1f TFLOP/s)" % (B, T, f, fl / f * 1e-6, b, 2.5 * fl / b * 1e-6))
