"""Is heads.LinearBiasFn (forward + backward) hipGraph-replay-safe when unrelated tensors are allocated after the capture?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import heads

torch.manual_seed(0)
for (R, K, N, bias) in ((32768, 768, 192, True), (32768, 192, 768, True), (32768, 192, 24, True), (65536, 96, 288, False), (8192, 384, 1536, True)):
    x = torch.randn(R, K, device="cuda").bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, device="cuda") * 0.05).requires_grad_(True)
    b = torch.randn(N, device="cuda").requires_grad_(True) if bias else None
    dy = torch.randn(R, N, device="cuda").bfloat16()

    def fb():
        for t in (x, w, b):
            if t is not None:
                t.grad = None
        y = heads.LinearBiasFn.apply(x, w, b, torch.bfloat16)
        y.backward(dy)
        return y

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fb()
        torch.cuda.synchronize()
        ref = (w.grad.clone(), x.grad.clone(), b.grad.clone() if bias else None)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            y = fb()
    torch.cuda.current_stream().wait_stream(side)
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (7, 96, 192, 384, 768, 1536, 96 * 96, 192 * 192, 384 * 384, 1 << 20, 1 << 22) for _ in range(20)]
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("R=%d K=%d N=%d replay %d: dW diff %.3e (|dW| %.3e)  dx diff %.3e  db diff %s" % (
            R, K, N, r, float((w.grad - ref[0]).abs().max()), float(ref[0].abs().max()), float((x.grad.float() - ref[1].float()).abs().max()),
            "%.3e" % float((b.grad - ref[2]).abs().max()) if bias else "-"))
    del junk
