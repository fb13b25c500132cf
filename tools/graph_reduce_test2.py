import torch, sys
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
xs = [torch.randn(1 << 20, device=dev) for _ in range(3)]
static = xs[0].clone()
def body():
    outs = []
    t = static
    for i in range(N):
        t = t * 1.0001 + 0.0001          # filler kernels
        if i % 3 == 0:
            outs.append((i, t.abs().mean(), None))
        elif i % 3 == 1:
            outs.append((i, t.view(-1, 128).sum(0), None))
        else:
            z = torch.zeros(4096, device=dev); z += t[:4096]
            outs.append((i, z.sum(), None))
    return outs
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = body()
for r in range(4):
    static.copy_(xs[r % 3])
    g.replay(); torch.cuda.synchronize()
    # eager reference
    t = xs[r % 3]; bad = 0; cat = [0, 0, 0]
    for i in range(N):
        t = t * 1.0001 + 0.0001
        if i % 3 == 0: ref = t.abs().mean()
        elif i % 3 == 1: ref = t.view(-1, 128).sum(0)
        else: ref = t[:4096].sum()
        got = outs[i][1]
        if not torch.allclose(got, ref, rtol=1e-4, atol=1e-4): bad += 1; cat[i % 3] += 1
    print("replay", r, "N", N, "bad outputs", bad, "by category [mean, colsum, zeros+add+sum]", cat)
