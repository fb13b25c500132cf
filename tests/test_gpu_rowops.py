"""-m gpu: the row-wise fused kernels (gm3d_amd/csrc/rowops.hip) against plain PyTorch fp32/fp64 references of the
same expressions (LayerNorm eps 1e-5, exact-erf GELU), forward and backward, fp32 (1e-5) and bf16 (bf16 rounding)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("adt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("R,T", [(25 * 8, 25), (64 * 37, 64), (3, 1), (5000, 1)])
def test_residual_ln_fwd_bwd(adt, tol, R, T):
    from gm3d_amd import fused
    torch.manual_seed(R)
    dev = "cuda"
    res = torch.randn(R, 384, device=dev)
    y = torch.randn(R, 384, device=dev).to(adt)
    add = torch.randn(R, 384, device=dev).to(adt)
    bias = torch.randn(384, device=dev) * 0.3
    gamma = 1 + 0.2 * torch.randn(384, device=dev)
    beta = 0.1 * torch.randn(384, device=dev)
    rs = (torch.rand(R // T, device=dev) > 0.3).float() / 0.7
    out_res, h, mean, rstd = fused.residual_ln_fwd(res, y, bias, rs, T, add, gamma, beta, 1e-5, adt, R)
    rsr = rs.repeat_interleave(T).unsqueeze(1).double()
    x64 = (res.double() + rsr * (y.double() + bias.double()) + add.double()).requires_grad_(True)
    h64 = F.layer_norm(x64, (384,), gamma.double(), beta.double(), 1e-5)
    assert rel(out_res, x64.detach()) <= 1e-6
    assert rel(h, h64.detach()) <= tol
    assert rel(mean, x64.detach().mean(1)) <= 1e-5 and rel(rstd, 1 / torch.sqrt(x64.detach().var(1, unbiased=False) + 1e-5)) <= 1e-5
    # optional inputs absent
    o2, h2, _, _ = fused.residual_ln_fwd(res, None, None, None, 1, None, gamma, beta, 1e-5, adt, R)
    assert torch.equal(o2, res) and rel(h2, F.layer_norm(res.double(), (384,), gamma.double(), beta.double(), 1e-5)) <= tol

    dh = torch.randn(R, 384, device=dev).to(adt)
    gin = torch.randn(R, 384, device=dev)
    acc = torch.ones(R, 384, device=dev)
    dx, dy, sums = fused.residual_ln_bwd(dh, gin, out_res, mean, rstd, gamma, rs, T, acc, True, adt, R)
    gam64 = gamma.double().requires_grad_(True)
    bet64 = beta.double().requires_grad_(True)
    x64b = out_res.double().requires_grad_(True)
    hh = F.layer_norm(x64b, (384,), gam64, bet64, 1e-5)
    hh.backward(dh.double())
    dx_ref = x64b.grad + gin.double()
    assert rel(dx, dx_ref) <= 2e-5
    assert rel(acc, dx_ref + 1) <= 2e-5
    assert rel(dy, rsr * dx_ref) <= tol
    assert rel(sums[0], gam64.grad) <= 2e-5 and rel(sums[1], bet64.grad) <= 2e-5
    assert rel(sums[2], (rsr * dx_ref).sum(0)) <= 2e-5      # summed in fp32 BEFORE dy is rounded to bf16


@pytest.mark.parametrize("adt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1.2e-2)])
@pytest.mark.parametrize("R,C", [(200, 1536), (8192, 1536), (7, 128)])
def test_bias_gelu_fwd_bwd(adt, tol, R, C):
    from gm3d_amd import fused
    torch.manual_seed(R + C)
    f = (torch.randn(R, C, device="cuda") * 2).to(adt)
    bias = torch.randn(C, device="cuda") * 0.5
    dg = torch.randn(R, C, device="cuda").to(adt)
    g = fused.bias_gelu_fwd(f, bias, adt)
    pre = (f.double() + bias.double()).requires_grad_(True)
    ref = F.gelu(pre)
    assert rel(g, ref.detach()) <= tol
    df, db = fused.bias_gelu_bwd(dg, f, bias, adt)
    ref.backward(dg.double())
    assert rel(df, pre.grad) <= tol
    assert rel(db, pre.grad.sum(0)) <= 2e-5             # summed in fp32 BEFORE df is rounded to bf16


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("B,T,nblk,train", [(4, 25, 3, True), (3, 64, 2, False)])
def test_fused_stack_matches_per_op_modules(bf16, B, T, nblk, train):
    """Whole fused stack (forward + every gradient) vs the per-op PyTorch modules of the same package."""
    from gm3d_amd import models_mae_learn_loss as M
    torch.manual_seed(7)
    dpr = [0.0, 0.2, 0.3][:nblk]
    dec = M.TransformerDecoder(embed_dim=384, depth=nblk, drop_path_rate=dpr, num_heads=6).cuda().train(train)
    for p in dec.parameters():
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    x = torch.randn(B, T, 384, device="cuda")
    pos = torch.randn(B, T, 384, device="cuda")
    w = torch.randn(B, T, 384, device="cuda")
    draws = {}

    def scale(Bn, p, training, device):
        if p == 0.0 or not training:
            return None
        k = len(draws.setdefault("s", []))
        g = torch.Generator().manual_seed(100 + k)
        m = ((1 - p) + torch.rand(Bn, generator=g)).floor_()
        draws["s"].append(m)
        return (m / (1 - p)).to(device)

    def dpath(t, p, training):
        if p == 0.0 or not training:
            return t
        k = len(draws.setdefault("d", []))
        g = torch.Generator().manual_seed(100 + k)
        m = ((1 - p) + torch.rand(t.shape[0], generator=g)).floor_().to(t.device, t.dtype)
        draws["d"].append(m)
        return t.div(1 - p) * m.reshape(-1, 1, 1)

    res = {}
    for fused_on in (False, True):
        M.FUSED_STACK = fused_on
        M.drop_path_scale, M.drop_path = scale, dpath
        draws.clear()
        xs, ps = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        dec.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            out = dec(xs, ps, 0)
        (out.float() * w).sum().backward()
        res[fused_on] = (out.detach().float(), xs.grad.clone(), ps.grad.clone(),
                         {k: p.grad.clone() for k, p in dec.named_parameters()})
    M.FUSED_STACK = True
    import importlib
    importlib.reload(M)
    tol = 3e-2 if bf16 else 2e-5
    assert rel(res[True][0], res[False][0]) <= tol
    assert rel(res[True][1], res[False][1]) <= tol and rel(res[True][2], res[False][2]) <= tol
    for k in res[False][3]:
        assert rel(res[True][3][k], res[False][3][k]) <= tol, k


@pytest.mark.parametrize("C", [96, 192, 384, 4, 512, 100])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R", [1, 7, 4100])
def test_plain_layernorm_any_width(R, C, dtype):
    """gm3d_ln_plain_fwd/bwd (heads.LayerNormFn) against torch.nn.functional.layer_norm in fp64 on the same (rounded) inputs."""
    from gm3d_amd import heads
    g = torch.Generator().manual_seed(R * 1000 + C)
    x = (torch.randn(R, C, generator=g) * 1.5 + 0.3).cuda().to(dtype).requires_grad_(True)
    w = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
    b = (torch.randn(C, generator=g) * 0.2).cuda().requires_grad_(True)
    dy = torch.randn(R, C, generator=g).cuda().to(dtype)
    assert heads.layer_norm_supported(x, C)
    h = heads.LayerNormFn.apply(x, w, b, 1e-5, dtype)
    h.backward(dy)
    x64 = x.detach().double().requires_grad_(True)
    w64, b64 = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x64, (C,), w64, b64, 1e-5)
    ref.backward(dy.double())
    tol = 2e-6 if dtype == torch.float32 else 2.0 ** -8
    assert float((h.double() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert float((x.grad.double() - x64.grad).abs().max()) <= tol * max(1.0, float(x64.grad.abs().max()))
    assert float((w.grad.double() - w64.grad).abs().max()) <= 2e-5 * max(1.0, float(w64.grad.abs().max()))
    assert float((b.grad.double() - b64.grad).abs().max()) <= 2e-5 * max(1.0, float(b64.grad.abs().max()))


@pytest.mark.parametrize("adt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C", [96, 192, 384, 100, 512])
@pytest.mark.parametrize("form", ["plain", "xz", "xy", "xybias_scale", "all"])
def test_add_layer_norm_against_fp64(adt, tol, C, form):
    """heads.AddLayerNormFn: s = x + rowscale[sample] * (y + ybias) + z, h = LayerNorm(s), forward and every gradient (x, y, ybias, z,
    gamma, beta) against the same expression in fp64 -- the pre-norm residual sums of the hierarchical encoder's blocks (widths
    96 / 192 / 384, Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:60-75) and odd widths."""
    from gm3d_amd import heads
    B, T = 6, 37
    g = torch.Generator(device="cuda").manual_seed(C + len(form))
    rnd = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    x = rnd(B, T, C).to(adt).requires_grad_(True)
    y = rnd(B, T, C).to(adt).requires_grad_(True) if form in ("xy", "xybias_scale", "all") else None
    z = rnd(B, T, C).to(adt).requires_grad_(True) if form in ("xz", "all") else None
    yb = (rnd(C) * 0.5).requires_grad_(True) if form in ("xybias_scale", "all") else None
    rs = (torch.rand(B, device="cuda", generator=g) > 0.3).float() / 0.7 if form in ("xybias_scale", "all") else None
    w = (1 + 0.2 * rnd(C)).requires_grad_(True)
    b = (0.2 * rnd(C)).requires_grad_(True)
    s, h = heads.AddLayerNormFn.apply(x, y, yb, rs, z, w, b, 1e-5, adt)
    gs, gh = rnd(B, T, C).to(adt), rnd(B, T, C).to(adt)
    (s.float() * gs.float()).sum().add((h.float() * gh.float()).sum()).backward()
    d = lambda t: None if t is None else t.detach().double().requires_grad_(True)
    x6, y6, z6, yb6, w6, b6 = d(x), d(y), d(z), d(yb), d(w), d(b)
    s6 = x6
    if y6 is not None:
        t = y6 + (yb6 if yb6 is not None else 0.0)
        s6 = s6 + (rs.double().view(B, 1, 1) * t if rs is not None else t)
    if z6 is not None:
        s6 = s6 + z6
    s_st = s6.detach().to(adt).double() + (s6 - s6.detach())          # normalised AS STORED (value of the rounded sum, gradient of the exact one)
    h6 = torch.nn.functional.layer_norm(s_st, (C,), w6, b6, 1e-5)
    ((s6 * gs.double()).sum() + (h6 * gh.double()).sum()).backward()
    rel = lambda a, r: float((a.double() - r).abs().max()) / max(float(r.abs().max()), 1e-12)
    assert rel(s, s6) <= tol and rel(h, h6) <= tol
    for name, a, r in (("x", x, x6), ("y", y, y6), ("z", z, z6), ("ybias", yb, yb6), ("gamma", w, w6), ("beta", b, b6)):
        if a is not None:
            assert rel(a.grad, r.grad) <= (tol if adt == torch.float32 else 3e-2), (name, rel(a.grad, r.grad))


@pytest.mark.parametrize("adt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_bias_gelu_fn(adt, tol):
    from gm3d_amd import heads
    g = torch.Generator(device="cuda").manual_seed(3)
    f = torch.randn(300, 768, device="cuda", generator=g).to(adt).requires_grad_(True)
    bias = torch.randn(768, device="cuda", generator=g).requires_grad_(True)
    out = heads.BiasGeluFn.apply(f, bias, adt)
    go = torch.randn(300, 768, device="cuda", generator=g).to(adt)
    out.backward(go)
    f6, b6 = f.detach().double().requires_grad_(True), bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.gelu(f6 + b6)
    ref.backward(go.double())
    rel = lambda a, r: float((a.double() - r).abs().max()) / float(r.abs().max())
    assert rel(out, ref) <= tol and rel(f.grad, f6.grad) <= tol and rel(bias.grad, b6.grad) <= tol
