"""Parity of the HIP kernels (through the C ABI) against the CPU oracle.  -m gpu only.

Bar: FPS / KNN / Chamfer-argmin indices bit-exact; Chamfer distances bit-exact (same fp32
expression); Chamfer gradients and f32 attention within 1e-5 relative; bf16 attention within
bf16 rounding of an fp32 torch reference (tolerance stated at the assert).
"""
import pytest
import torch

from tests import clouds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gops():
    from gm3d_amd import ops
    return ops


def dev(t):
    return t.cuda().contiguous()


@pytest.mark.parametrize("family", list(clouds.FAMILIES))
@pytest.mark.parametrize("N,G", [(1024, 64), (100, 17), (2048, 128), (300, 300), (8192, 1200), (4096, 512), (5000, 100), (2048, 512),
                                 (12000, 64)])
def test_fps_index_exact(gops, oracle_ops, family, N, G):
    x = clouds.FAMILIES[family](3, N, seed=11)
    ref = oracle_ops.furthest_point_sample(x, G)
    idx, cen = gops.fps(dev(x), G)
    assert idx.dtype == torch.int32 and idx.shape == (3, G)
    assert torch.equal(idx.cpu(), ref)
    exp_cen = torch.gather(x, 1, ref.long().unsqueeze(-1).expand(-1, -1, 3))
    assert torch.equal(cen.cpu(), exp_cen)
    # reference-side API: furthest_point_sample + gather_operation (models_mae_learn_loss.py:931-932)
    idx2 = gops.furthest_point_sample(dev(x), G)
    got = gops.gather_operation(dev(x).transpose(1, 2).contiguous(), idx2).transpose(1, 2).contiguous()
    assert torch.equal(got.cpu(), exp_cen)


def test_fps_large_cloud(gops, oracle_ops):
    x = clouds.uniform(1, 8192, seed=5)
    ref = oracle_ops.furthest_point_sample(x, 1024)
    assert torch.equal(gops.furthest_point_sample(dev(x), 1024).cpu(), ref)
    x = clouds.gaussian(1, 16384, seed=6)
    ref = oracle_ops.furthest_point_sample(x, 64)
    assert torch.equal(gops.furthest_point_sample(dev(x), 64).cpu(), ref)


def test_fps_matches_reference_numpy_fps(gops):
    """gm3d_fps against the ONE statement of FPS the reference tree holds (P/datasets/ModelNetDataset.py:25-46, run in place by
    tests/golden/make_golden_fps.py: start rotated to index 0, clouds clear of the 1e-3 ball).  No oracle involved: the
    fixture holds the reference's own output.  Cases: 1024->64 (pretrain grouping), 8192->1024 (SVM validation), 8192->1200
    (fine-tune), and the `_tight` ones, which only the reference's float32 arithmetic reproduces (its float64 run differs)."""
    import os
    import numpy as np
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "fps_modelnet.npz"))
    names = sorted({k.split("/")[0] for k in fx.files})
    assert len(names) >= 5
    for n in names:
        xyz, want, pts = torch.from_numpy(fx[n + "/xyz"]), torch.from_numpy(fx[n + "/idx"]), torch.from_numpy(fx[n + "/points"])
        idx, cen = gops.fps(dev(xyz[None]), len(want))
        assert torch.equal(idx.cpu()[0], want), n
        assert torch.equal(cen.cpu()[0], pts), n
    # all cases of one size as ONE batch (the kernel's per-cloud workgroups must not interact)
    big = [n for n in names if fx[n + "/xyz"].shape[0] == 8192]
    x = torch.stack([torch.from_numpy(fx[n + "/xyz"]) for n in big])
    idx, _ = gops.fps(dev(x), 1024)
    for i, n in enumerate(big):
        assert torch.equal(idx.cpu()[i], torch.from_numpy(fx[n + "/idx"][:1024])), n      # FPS is prefix-stable


@pytest.mark.parametrize("family", ["uniform", "gaussian", "lattice", "duplicates"])
@pytest.mark.parametrize("N,G,k", [(1024, 64, 32), (1024, 128, 16), (1024, 256, 8), (100, 7, 64), (2048, 33, 5),
                                   (1000, 10, 1), (777, 5, 32), (64, 3, 64)])
def test_knn_index_exact(gops, oracle_ops, family, N, G, k):
    x = clouds.FAMILIES[family](2, N, seed=3)
    fidx = oracle_ops.furthest_point_sample(x, G)
    center = torch.gather(x, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    rd, ri = oracle_ops.knn(x, center, k)
    dist, idx = gops.KNN(k=k, transpose_mode=True)(dev(x), dev(center))
    assert idx.dtype == torch.int64
    assert torch.equal(idx.cpu(), ri)
    assert torch.equal(dist.cpu(), rd)  # sqrt of an identical fp32 value, correctly rounded on both sides
    rnb, rnbo = oracle_ops.group(x, center, ri)
    nb, nbo, idx2 = gops.knn_group(dev(x), dev(center), k)
    assert torch.equal(idx2.cpu(), ri)
    assert torch.equal(nb.cpu(), rnb) and torch.equal(nbo.cpu(), rnbo)


def test_knn_big_cloud(gops, oracle_ops):
    x = clouds.uniform(1, 8192, seed=9)
    q = x[:, :40].contiguous()
    rd, ri = oracle_ops.knn(x, q, 32)
    dist, idx = gops.knn(dev(x), dev(q), 32)
    assert torch.equal(idx.cpu(), ri)


@pytest.mark.parametrize("P,n,m", [(4992, 32, 32), (7, 32, 32), (5, 50, 70), (2, 1500, 1100), (3, 1, 9), (70001, 8, 8), (37, 8, 8), (4099, 16, 16)])
def test_chamfer_forward_backward(gops, oracle_ops, P, n, m):
    g = torch.Generator().manual_seed(P * 131 + n)
    a = (torch.rand(P, n, 3, generator=g) - 0.5)
    b = (torch.rand(P, m, 3, generator=g) - 0.5)
    if n == m:  # exact duplicates across the two clouds and inside one cloud: argmin ties
        b[:, :3] = a[:, :3]
        a[:, 5 % n] = a[:, 4 % n]
    ga1 = torch.randn(P, n, generator=g)
    ga2 = torch.randn(P, m, generator=g)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    d1r, d2r, i1r, i2r = oracle_ops.chamfer(ar, br)
    ((d1r * ga1).sum() + (d2r * ga2).sum()).backward()
    ag, bg = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    d1, d2, i1, i2 = gops.chamfer(ag, bg)
    assert torch.equal(i1.cpu(), i1r) and torch.equal(i2.cpu(), i2r)
    assert torch.equal(d1.cpu(), d1r.detach()) and torch.equal(d2.cpu(), d2r.detach())
    ((d1 * ga1.cuda()).sum() + (d2 * ga2.cuda()).sum()).backward()
    # fp32 accumulation order differs from the oracle's fp64 sums: 1e-5 relative to the gradient scale
    for got, ref in ((ag.grad.cpu(), ar.grad), (bg.grad.cpu(), br.grad)):
        scale = ref.abs().max().clamp_min(1e-6)
        assert (got - ref).abs().max() <= 1e-5 * scale


def test_chamfer_modules(gops, oracle_ops):
    g = torch.Generator().manual_seed(1)
    a, b = torch.rand(64, 32, 3, generator=g), torch.rand(64, 32, 3, generator=g)
    per_point = gops.ChamferDistanceL2()(dev(a), dev(b))
    assert per_point.shape == (64, 32)
    assert torch.equal(per_point.cpu(), oracle_ops.ChamferDistanceL2()(a, b))
    mean = gops.ChamferDistanceL2(reduction="mean")(dev(a), dev(b))
    assert abs(mean.item() - oracle_ops.ChamferDistanceL2("mean")(a, b).item()) <= 1e-6
    l1 = gops.ChamferDistanceL1()(dev(a), dev(b))
    assert abs(l1.item() - oracle_ops.ChamferDistanceL1()(a, b).item()) <= 1e-6


def _attn_ref(qkv, H, scale):
    B, T, _ = qkv.shape
    q, k, v = qkv.reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)  # models/Point_MAE.py:115-116
    attn = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(B, T, H * 64)


@pytest.mark.parametrize("T", [64, 25, 39, 1, 32, 33, 65, 96, 97, 128])
def test_attention_f32(gops, T):
    torch.manual_seed(T)
    B, H = 5, 6
    qkv = (torch.randn(B, T, 3 * H * 64, device="cuda") * 1.5).requires_grad_(True)
    w = torch.randn(B, T, H * 64, device="cuda")
    ref = _attn_ref(qkv.double(), H, 0.125)
    (ref * w.double()).sum().backward()
    gref = qkv.grad.clone()
    qkv.grad = None
    out = gops.attention(qkv, H, 0.125)
    (out * w).sum().backward()
    assert (out.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert (qkv.grad.double() - gref.double()).abs().max() <= 1e-5 * gref.abs().max()


@pytest.mark.parametrize("T", [64, 25, 39, 65, 128])
def test_attention_bf16(gops, T):
    torch.manual_seed(100 + T)
    B, H = 4, 6
    qkv32 = torch.randn(B, T, 3 * H * 64, device="cuda")
    qkv = qkv32.bfloat16().requires_grad_(True)
    w = torch.randn(B, T, H * 64, device="cuda").bfloat16()
    q64 = qkv.detach().double().requires_grad_(True)
    ref = _attn_ref(q64, H, 0.125)
    (ref * w.double()).sum().backward()
    out = gops.attention(qkv, H, 0.125)
    (out.float() * w.float()).sum().backward()
    # bf16 has 8 significant bits: P and the outputs are rounded to bf16 -> 2e-2 of the tensor scale
    assert (out.double() - ref).abs().max() <= 2e-2 * ref.abs().max()
    assert (qkv.grad.double() - q64.grad).abs().max() <= 3e-2 * q64.grad.abs().max()


def test_errors_are_loud(gops):
    with pytest.raises(RuntimeError):
        gops.fps(torch.zeros(1, 16, 3), 4)  # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        gops.fps(torch.zeros(1, 3, 16, device="cuda").transpose(1, 2), 4)  # non-contiguous like upstream
    with pytest.raises(RuntimeError):
        gops.knn(torch.zeros(1, 8, 3, device="cuda"), torch.zeros(1, 2, 3, device="cuda"), 9)  # k > N


def test_scale_translate_kernel_equals_op_chain():
    """The augmentation as one launch (gm3d_scale_translate) against the elementwise-op chain with the same uniform draws."""
    from gm3d_amd.engine_pretrain import PointcloudScaleAndTranslate
    aug = PointcloudScaleAndTranslate()
    pc = torch.randn(7, 1024, 3, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(11)
    got = aug(pc.clone(), generator=g)
    g = torch.Generator(device="cuda").manual_seed(11)
    u = torch.rand(2, 7, 3, device="cuda", generator=g)
    scale = u[0] * (aug.scale_high - aug.scale_low) + aug.scale_low
    shift = (u[1] * 2 - 1) * aug.translate_range
    want = aug(pc.clone(), draws=(scale, shift))
    assert torch.equal(got, want)
    assert float(scale.min()) >= 2.0 / 3.0 - 1e-6 and float(scale.max()) <= 1.5 + 1e-6 and float(shift.abs().max()) <= 0.2 + 1e-6


@pytest.mark.parametrize("n", [8, 16])
def test_chamfer_small_patch_backward_is_the_sequential_sum(gops, n):
    """n == m == 8 / 16 (Point-M2AE's fine patches): the backward has no atomics; every slot is summed in the order of the sequential loops
    (direction 1 over ascending i, then direction 2 over ascending j), so it EQUALS an fp32 restatement of those loops."""
    import numpy as np
    P = 41
    g = torch.Generator().manual_seed(n)
    a = torch.rand(P, n, 3, generator=g) - 0.5
    b = torch.rand(P, n, 3, generator=g) - 0.5
    b[:, :3] = a[:, :3]                      # ties and many-to-one nearest neighbours
    b[:, 5] = b[:, 4]
    w1, w2 = torch.randn(P, n, generator=g), torch.randn(P, n, generator=g)
    ag, bg = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
    d1, d2, i1, i2 = gops.chamfer(ag, bg)
    ((d1 * w1.cuda()).sum() + (d2 * w2.cuda()).sum()).backward()
    A, B, I1, I2 = a.numpy(), b.numpy(), i1.cpu().numpy(), i2.cpu().numpy()
    W1, W2 = w1.numpy(), w2.numpy()
    ga, gb = np.zeros_like(A), np.zeros_like(B)
    f = np.float32
    for p in range(P):
        for i in range(n):
            t = f(f(f(2.0) * f(A[p, i] - B[p, I1[p, i]])) * W1[p, i])
            ga[p, i] = f(ga[p, i] + t)
            gb[p, I1[p, i]] = f(gb[p, I1[p, i]] - t)
        for j in range(n):
            t = f(f(f(2.0) * f(B[p, j] - A[p, I2[p, j]])) * W2[p, j])
            gb[p, j] = f(gb[p, j] + t)
            ga[p, I2[p, j]] = f(ga[p, I2[p, j]] - t)
    assert np.array_equal(ag.grad.cpu().numpy(), ga) and np.array_equal(bg.grad.cpu().numpy(), gb)


def test_chamfer_general_backward_survives_graph_replay(gops):
    """The general-shape Chamfer backward (any patch size without a kernel of its own) zero-fills its outputs before an atomic scatter.  With
    hipMemsetAsync as the fill, a captured graph came back with 1e34-sized garbage once the process had made other host-to-device
    copies after the capture (found by tools/m2ae_graph_diag.py): the fill is a kernel of the C ABI now.  Replay after such copies
    must equal the eager result bit for bit (the atomics add one term per slot and direction: order cannot matter)."""
    P, n = 4096, 12
    g = torch.Generator().manual_seed(3)
    a = torch.randn(P, n, 3, generator=g).cuda().requires_grad_(True)
    b = torch.randn(P, n, 3, generator=g).cuda()
    w = torch.rand(P, n, generator=g).cuda()

    def fb():
        a.grad = None
        d1, d2, _, _ = gops.chamfer(a, b)
        ((d1 + d2) * w).sum().backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fb()
        fb()
        torch.cuda.synchronize()
        want = a.grad.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            fb()
    torch.cuda.current_stream().wait_stream(side)
    junk = [torch.nn.Linear(384, 1536).cuda() for _ in range(40)]           # host-to-device copies + allocations after the capture
    junk += [torch.full((1 << 18,), float("nan"), device="cuda") for _ in range(16)]
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(a.grad, want)
    del junk


@pytest.mark.parametrize("R,C", [(8192, 1536), (300, 2048), (4096, 24), (65536, 96)])
def test_colsum_wide(R, C):
    """gm3d_colsum_partial + finish up to 2048 columns (the bias gradients of Point-M2AE's Linear layers) against an fp64 sum."""
    from gm3d_amd import embed
    for dt in (torch.bfloat16, torch.float32):
        m = torch.randn(R, C, generator=torch.Generator().manual_seed(R + C)).cuda().to(dt)
        got = embed.colsum(m, dt)
        want = m.double().sum(0)
        assert float((got.double() - want).abs().max()) <= 2e-6 * float(m.double().abs().sum(0).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T", [129, 256, 257, 512])
def test_attention_beyond_128_tokens(gops, dtype, T):
    """ops.attention for 128 < T <= 512 (cfgs/config_3.yaml: 256 groups of 8; cls + 256 in fine-tuning): routed to the flash-style
    kernel of the hierarchical encoder with a NULL mask.  Forward and input gradient against an fp64 torch softmax attention;
    fp32: 1e-5, bf16: one rounding of the operands and of the output (2e-2 / 3e-2 of the tensor scale)."""
    B, H, hd = 3, 6, 64
    if dtype == torch.bfloat16 and T > 256:
        # head_dim 64 in bf16: the backward keeps q, k, v, dO of the head in LDS (4 x T x 144 B): T <= 256; the C ABI says so
        from gm3d_amd._capi import Gm3dError
        with pytest.raises(Gm3dError):
            q = torch.zeros(1, T, 3 * H * hd, device="cuda", dtype=dtype, requires_grad=True)
            gops.attention(q, H, hd ** -0.5).sum().backward()
        return
    g = torch.Generator(device="cuda").manual_seed(T)
    qkv = (torch.randn(B, T, 3 * H * hd, device="cuda", generator=g) * 0.7).to(dtype).requires_grad_(True)
    dout = torch.randn(B, T, H * hd, device="cuda", generator=g).to(dtype)
    out = gops.attention(qkv, H, hd ** -0.5)
    out.backward(dout)
    q64 = qkv.detach().double().requires_grad_(True)
    q, k, v = q64.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    ref = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B, T, H * hd)
    ref.backward(dout.double())
    tol_o, tol_g = (1e-5, 1e-5) if dtype == torch.float32 else (2e-2, 3e-2)
    assert float((out.double() - ref).abs().max()) <= tol_o * float(ref.abs().max())
    assert float((qkv.grad.double() - q64.grad).abs().max()) <= tol_g * float(q64.grad.abs().max())
