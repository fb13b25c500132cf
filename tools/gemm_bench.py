"""Times the step's GEMM shapes through torch (hipBLASLt) on the GPU: F.linear (NT), dgrad (NN), wgrad (TN)."""
import sys, torch
torch.manual_seed(0)
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [("qkv", 8192, 384, 1152), ("proj", 8192, 384, 384), ("fc1", 8192, 384, 1536), ("fc2", 8192, 1536, 384),
          ("qkv25", 3200, 384, 1152), ("fc1_25", 3200, 384, 1536), ("fc2_25", 3200, 1536, 384),
          ("conv2", 262144, 128, 256), ("conv3l", 262144, 256, 512), ("conv4", 262144, 512, 384)]
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * K * N
    a = t(lambda: x @ w.t())
    b = t(lambda: dy @ w)
    c = t(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
    c2 = t(lambda: dy.t() @ x)
    print("%-7s M=%6d K=%4d N=%4d  fwd %7.1f us (%5.0f TF)  dgrad %7.1f us (%5.0f TF)  wgrad f32-out %7.1f us (%5.0f TF)  wgrad bf16-out %7.1f us"
          % (name, M, K, N, a, fl / a / 1e6, b, fl / b / 1e6, c, fl / c / 1e6, c2))
# sliced weight (the mini-PointNet concat trick) and batched wgrad
x = torch.randn(262144, 256, device=dev, dtype=torch.bfloat16); w = torch.randn(512, 512, device=dev, dtype=torch.bfloat16)
print("conv3 with w[:,256:] view: %.1f us; contiguous: %.1f us" % (t(lambda: x @ w[:, 256:].t()), t(lambda: x @ w[:, 256:].contiguous().t())))
for nb in (4, 8, 12):
    dy = torch.randn(nb, 8192, 1536, device=dev, dtype=torch.bfloat16); xx = torch.randn(nb, 8192, 384, device=dev, dtype=torch.bfloat16)
    tt = t(lambda: torch.bmm(dy.transpose(1, 2), xx))
    print("bmm wgrad fc1 x%d: %.1f us total, %.1f us each (%.0f TF)" % (nb, tt, tt / nb, nb * 2.0 * 8192 * 1536 * 384 / tt / 1e6))
    try:
        tt = t(lambda: torch.bmm(dy.transpose(1, 2), xx, out_dtype=torch.float32))
        print("   f32-out: %.1f us total" % tt)
    except Exception as ex:
        print("   bmm out_dtype unsupported:", type(ex).__name__)
