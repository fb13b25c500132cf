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
