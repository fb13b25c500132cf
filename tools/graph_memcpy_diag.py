"""Are memcpy / memset NODES of a captured hipGraph replay-safe on this stack when the process makes other copies after the capture?
(The memset case is not: tools/m2ae_graph_diag.py)   python tools/graph_memcpy_diag.py"""
import torch

dev = "cuda"
torch.manual_seed(0)
for what in ("copy_ (D2D memcpy node)", "zero_ ", "torch.zeros", "hipMemsetAsync via Tensor.fill_(0) on bytes"):
    bad = 0
    for n in (96, 4096, 1 << 16, 1 << 20, 3 << 20, 25165824 // 4):
        x = torch.randn(n, device=dev)
        y = torch.empty(n, device=dev)
        z = torch.empty(n, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                if what.startswith("copy_"):
                    y.copy_(x)
                    z = y * 2.0
                elif what.startswith("zero_"):
                    y.copy_(x)
                    y.zero_()
                    z = y + x
                elif what.startswith("torch.zeros"):
                    t = torch.zeros(n, device=dev)
                    z = t + x
                else:
                    yb = y.view(torch.uint8)
                    yb.fill_(0)
                    z = y + x
        torch.cuda.current_stream().wait_stream(side)
        junk = [torch.nn.Linear(384, 1536).cuda() for _ in range(40)] + [torch.randn(1 << 20).cuda() for _ in range(8)]
        for r in range(3):
            x.copy_(torch.randn(n, device=dev))
            y.fill_(float("nan")) if not what.startswith("copy_") else None
            g.replay()
            torch.cuda.synchronize()
            want = x * 2.0 if what.startswith("copy_") else x
            if not torch.equal(z, want):
                bad += 1
        del junk
    print("%-45s wrong replays: %d of 18" % (what, bad))
