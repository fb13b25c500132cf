"""-m gpu: the fused mini-PointNet embed (gm3d_amd/embed.py + csrc/embed.hip) against the per-op PyTorch modules of the
same package (which tests/test_gpu_model.py ties to the reference fixtures): tokens, every parameter gradient and the
BatchNorm running statistics, train and eval mode, fp32 (1e-5) and bf16 (bf16 rounding)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b, floor=1e-12):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


def _run(M, enc, nb, w, fused, bf16, train):
    M.FUSED_EMBED = fused
    enc.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        if train:
            tok = enc(nb)
            (tok.float() * w).sum().backward()
        else:
            with torch.no_grad():
                tok = enc(nb)
    M.FUSED_EMBED = True
    grads = {k: p.grad.detach().double().clone() for k, p in enc.named_parameters()} if train else {}
    bufs = {k: b.detach().double().clone() for k, b in enc.named_buffers()}
    return tok.detach().double(), grads, bufs


def _err(a, b):
    return float((a - b).abs().max())


@pytest.mark.parametrize("B,G,train", [(2, 64, True), (3, 16, True), (2, 64, False)])
def test_embed_matches_modules(B, G, train):
    """fp32: fused == per-op modules to 1e-5.  bf16: both are bf16 approximations of the fp32 result with different
    rounding points (BatchNorm backward cancels large sums), so the fused path must be as close to fp32 as the
    autocast module path is (x2 slack)."""
    from gm3d_amd import models_mae_learn_loss as M
    torch.manual_seed(B * 100 + G)
    base = M.Encoder(384).cuda()
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
    nb = (torch.randn(B, G, 32, 3, device="cuda") * 0.2)
    w = torch.randn(B, G, 384, device="cuda")
    run = lambda fused, bf16: _run(M, copy.deepcopy(base).train(train), nb, w, fused, bf16, train)
    ref_tok, ref_g, ref_b = run(False, False)
    tok, g, b = run(True, False)
    assert _err(tok, ref_tok) <= 1e-5 * float(ref_tok.abs().max())
    gnorm = sum(float(v.pow(2).sum()) for v in ref_g.values()) ** 0.5
    for k in ref_g:   # biases in front of a BatchNorm: analytically zero (exact 0 here, rounding noise in autograd)
        assert _err(g[k], ref_g[k]) <= 3e-5 * float(ref_g[k].abs().max()) + 1e-6 * gnorm, k
    for k in ref_b:
        assert _err(b[k], ref_b[k]) <= 1e-5 * float(ref_b[k].abs().max()) + 1e-7, k
    mtok, mg, mb = run(False, True)      # autocast modules
    ftok, fg, fb = run(True, True)       # fused bf16
    assert _err(ftok, ref_tok) <= 2 * _err(mtok, ref_tok) + 2e-2 * float(ref_tok.abs().max())   # bf16: 8 significant bits
    for k in ref_g:
        assert _err(fg[k], ref_g[k]) <= 2 * _err(mg[k], ref_g[k]) + 1e-3 * float(ref_g[k].abs().max()) + 1e-6 * gnorm, k
    for k in ref_b:
        assert _err(fb[k], ref_b[k]) <= 2 * _err(mb[k], ref_b[k]) + 1e-3 * float(ref_b[k].abs().max()) + 1e-6, k


def test_splitk_wgrad_and_colsum():
    from gm3d_amd import embed
    torch.manual_seed(0)
    for adt in (torch.float32, torch.bfloat16):
        dy = torch.randn(8192, 96, device="cuda").to(adt)
        x = torch.randn(8192, 40, device="cuda").to(adt)
        ref = dy.double().t() @ x.double()
        assert rel(embed.splitk_wgrad(dy, x), ref) <= 1e-5
        assert rel(embed.colsum(dy.contiguous(), adt), dy.double().sum(0)) <= 1e-5


@pytest.mark.parametrize("B,G,V,bf16", [(4, 64, 25, False), (4, 64, 25, True), (3, 16, 5, False), (2, 64, 64, False)])
def test_visible_only_embed(B, G, V, bf16):
    """Encoder(nb, vis_ids) -- BN-apply, the last conv and its max-pool on the visible groups only -- against
    take(Encoder(nb), vis_ids): tokens (the same dot products; a library GEMM may tile the smaller matrix differently), every parameter gradient (the dropped rows'
    gradient is exactly zero; only the summation order of the weight gradient differs) and the running statistics (still those
    of the full batch).  vis_ids is a strided view of an (B,L) order buffer, unsorted, as the mask kernel produces it."""
    from gm3d_amd import models_mae_learn_loss as M
    from gm3d_amd.models_mae_learn_loss import take
    torch.manual_seed(7 + V)
    base = M.Encoder(384).cuda()
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    nb = torch.randn(B, G, 32, 3, device="cuda") * 0.2
    order = torch.stack([torch.randperm(G, device="cuda") for _ in range(B)])          # (B,G) int64
    vis = order[:, :V]
    assert vis.stride(0) == G
    w = torch.randn(B, V, 384, device="cuda")

    def run(selected):
        enc = copy.deepcopy(base).train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            tok = enc(nb, vis_ids=vis) if selected else take(enc(nb), vis)
            (tok.float() * w).sum().backward()
        return (tok.detach(), {k: p.grad.detach().double() for k, p in enc.named_parameters()},
                {k: b.detach().double().clone() for k, b in enc.named_buffers()})

    tok_f, g_f, b_f = run(False)
    tok_s, g_s, b_s = run(True)
    assert tok_s.shape == (B, V, 384)
    assert _err(tok_s.double(), tok_f.double()) <= (8e-3 if bf16 else 1e-5) * float(tok_f.abs().max())
    gnorm = sum(float(v.pow(2).sum()) for v in g_f.values()) ** 0.5
    tol = 2e-2 if bf16 else 2e-5        # bf16: da2 is rounded to bf16 in both runs, the weight-gradient partial sums differ
    for k in g_f:
        assert _err(g_s[k], g_f[k]) <= tol * float(g_f[k].abs().max()) + 1e-6 * gnorm, k
    for k in b_f:
        assert torch.equal(b_s[k], b_f[k]), k


def test_group_select_maps():
    from gm3d_amd._capi import check, lib
    from gm3d_amd.ops import _ptr, _stream
    B, G, V = 5, 64, 25
    order = torch.stack([torch.randperm(G, device="cuda") for _ in range(B)])
    sel = torch.empty(B * V, dtype=torch.int32, device="cuda")
    inv = torch.empty(B * G, dtype=torch.int32, device="cuda")
    check(lib.gm3d_group_select_maps(_ptr(order), G, B, V, G, _ptr(sel), _ptr(inv), _stream()), "maps")
    want_sel = (order[:, :V] + torch.arange(B, device="cuda")[:, None] * G).reshape(-1).int()
    assert torch.equal(sel, want_sel)
    want_inv = torch.full((B * G,), -1, dtype=torch.int32, device="cuda")
    want_inv[want_sel.long()] = torch.arange(B * V, dtype=torch.int32, device="cuda")
    assert torch.equal(inv, want_inv)
    assert lib.gm3d_group_select_maps(_ptr(order), V - 1, B, V, G, _ptr(sel), _ptr(inv), _stream()) != 0      # pitch < V


@pytest.mark.parametrize("bf16", [False, True])
def test_embed_at_point_m2ae_level0_sizes(bf16):
    """The fused node on Point-M2AE's level-0 token embed (gm3d_amd/point_m2ae.TokenEmbed(3, 96): 16-point groups, 512 groups per
    cloud, a 96-wide output -- max-pools outside the GEMM epilogue, narrow last conv) against the same module run op by op."""
    from gm3d_amd import embed, point_m2ae as P
    torch.manual_seed(5)
    base = P.TokenEmbed(3, 96).cuda().train()
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    nb = torch.randn(2, 512, 16, 3, device="cuda") * 0.2
    w = torch.randn(2, 512, 96, device="cuda")

    def run(fused, amp):
        enc = copy.deepcopy(base)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            tok = embed.run_embed(enc, nb) if fused else enc(nb)
            (tok.float() * w).sum().backward()
        return (tok.detach().double(), {k: p.grad.detach().double() for k, p in enc.named_parameters()},
                {k: b.detach().double() for k, b in enc.named_buffers()})

    ref_tok, ref_g, ref_b = run(False, False)
    gnorm = sum(float(v.pow(2).sum()) for v in ref_g.values()) ** 0.5
    tok, g, b = run(True, bf16)
    if not bf16:
        # 16,384 rows in front of every BatchNorm: the weight gradients there are small differences of large sums, and the fp32
        # op chain is itself only good to ~1e-4 of the largest entry -- the yardstick is the same module in fp64
        enc64 = copy.deepcopy(base).cpu().double()           # on the CPU: the module's plain F.linear / BatchNorm path
        tok64 = enc64(nb.cpu().double())
        (tok64 * w.cpu().double()).sum().backward()
        g64 = {k: p.grad.detach().cuda() for k, p in enc64.named_parameters()}
        tok64 = tok64.detach().cuda()
        assert _err(tok, tok64) <= 1e-5 * float(tok64.abs().max())
        for k in ref_g:
            e_f, e_m = _err(g[k], g64[k]), _err(ref_g[k], g64[k])
            assert e_f <= max(2.0 * e_m, 3e-5 * float(g64[k].abs().max()) + 1e-6 * gnorm), (k, e_f, e_m)
        for k in ref_b:
            assert _err(b[k], ref_b[k]) <= 1e-5 * float(ref_b[k].abs().max()) + 1e-7, k
    else:
        mtok, mg, mb = run(False, True)
        assert _err(tok, ref_tok) <= 2 * _err(mtok, ref_tok) + 2e-2 * float(ref_tok.abs().max())
        for k in ref_g:
            assert _err(g[k], ref_g[k]) <= 2 * _err(mg[k], ref_g[k]) + 1e-3 * float(ref_g[k].abs().max()) + 1e-6 * gnorm, k
        for k in ref_b:
            assert _err(b[k], ref_b[k]) <= 2 * _err(mb[k], ref_b[k]) + 1e-3 * float(ref_b[k].abs().max()) + 1e-6, k


@pytest.mark.parametrize("in_c,out_c,G,k", [(96, 192, 256, 8), (192, 384, 64, 8)])
@pytest.mark.parametrize("bf16", [False, True])
def test_deep_token_embed_fused_vs_module(in_c, out_c, G, k, bf16):
    """Point-M2AE's level-1 / level-2 token embeds (point_m2ae.TokenEmbed with token features as input) on heads.BnBcastActFn + split
    second-conv weights (TokenEmbed._forward_fused) against the plain module: tokens, every parameter gradient, the input gradient
    and the BatchNorm buffers.  fp32: as close to an fp64 CPU run as the fp32 op chain (x2); bf16: as close to fp32 as autocast (x2)."""
    from gm3d_amd import point_m2ae as P
    torch.manual_seed(in_c + G)
    base = P.TokenEmbed(in_c, out_c).cuda().train()
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    x = torch.randn(2, G, k, in_c, device="cuda") * 0.5
    w = torch.randn(2, G, out_c, device="cuda")

    def run(fused, amp):
        enc = copy.deepcopy(base)
        xi = x.clone().requires_grad_(True)
        P.FUSED_EMBED_DEEP = False
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            tok = enc._forward_fused(xi) if fused else enc(xi)
            (tok.float() * w).sum().backward()
        P.FUSED_EMBED_DEEP = True
        g = {k_: p.grad.detach().double() for k_, p in enc.named_parameters()}
        g["input"] = xi.grad.detach().double()
        return tok.detach().double(), g, {k_: b.detach().double() for k_, b in enc.named_buffers()}

    enc64 = copy.deepcopy(base).cpu().double()
    x64 = x.cpu().double().requires_grad_(True)
    tok64 = enc64(x64)
    (tok64 * w.cpu().double()).sum().backward()
    g64 = {k_: p.grad.detach().cuda() for k_, p in enc64.named_parameters()}
    g64["input"] = x64.grad.cuda()
    tok64 = tok64.detach().cuda()
    gnorm = sum(float(v.pow(2).sum()) for v in g64.values()) ** 0.5
    mtok, mg, mb = run(False, bf16)
    tok, g, b = run(True, bf16)
    tol_t, tol_g = (1e-5, 3e-5) if not bf16 else (2e-2, 1e-3)
    assert _err(tok, tok64) <= 2 * _err(mtok, tok64) + tol_t * float(tok64.abs().max())
    for k_ in g64:
        assert _err(g[k_], g64[k_]) <= 2 * _err(mg[k_], g64[k_]) + tol_g * float(g64[k_].abs().max()) + 1e-6 * gnorm, k_
    for k_ in mb:
        assert _err(b[k_], mb[k_]) <= (1e-5 if not bf16 else 1e-2) * float(mb[k_].abs().max()) + 1e-6, k_
