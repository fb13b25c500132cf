"""-m gpu: LayerNorm folded into the GEMMs around it (gm3d_gemm_tn_bf16_res / gm3d_gemm_tn_bf16_lna, fused.TransformerStackFn with
gemm.FUSE_LN) against the three-kernel form it replaces (GEMM -> gm3d_residual_ln_fwd -> GEMM): the residual stream must be
IDENTICAL (same products, same rounding points), the row statistics exact, the normalised rows within the stated bf16 bound.
The folded form is OFF in the product (measured 2.5 % slower on the step); these tests keep its kernels honest."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,with_add,with_scale", [(4096, 384, False, True), (3200, 1536, True, True), (8192, 1536, True, False),
                                                      (200, 384, True, True), (70, 1536, False, False)])
def test_residual_epilogue_and_layernorm_on_load(M, K, with_add, with_scale):
    from gm3d_amd import gemm, fused
    C, T = 384, 25 if M % 25 == 0 else (64 if M % 64 == 0 else 10)
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(C, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    res = torch.randn(M, C, device="cuda", generator=g)
    add = torch.randn(M, C, device="cuda", generator=g).bfloat16() if with_add else None
    rs = (torch.rand(M // T, device="cuda", generator=g) + 0.5) if with_scale else None
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g) * 0.1
    # reference: ring GEMM (bit-identical product) -> stand-alone residual LayerNorm
    y = gemm.linear_tn_ring(a, w)
    u_ref, h_ref, m_ref, r_ref = fused.residual_ln_fwd(res, y, bias, rs, T, add, gamma, beta, 1e-5, torch.bfloat16, M)
    U, U16, stats = gemm.linear_res(a, w, bias, res, rs, T, add)
    assert float((U - u_ref).abs().max()) <= 1e-6 * float(u_ref.abs().max())       # same sums, FMA contraction aside
    assert torch.equal(U16, U.bfloat16())
    # the per-tile statistics against torch
    Ut = U.view(M, 3, 128)
    assert torch.allclose(stats[:, :, 0].t(), Ut.mean(-1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[:, :, 1].t(), (Ut - Ut.mean(-1, keepdim=True)).pow(2).sum(-1), rtol=1e-4, atol=1e-5)
    # consumer: qkv-shaped plain GEMM and fc1-shaped GELU GEMM with the LayerNorm applied on load
    wq = (torch.randn(1152, C, device="cuda", generator=g) / C ** 0.5).bfloat16()
    w1 = (torch.randn(1536, C, device="cuda", generator=g) / C ** 0.5).bfloat16()
    b1 = torch.randn(1536, device="cuda", generator=g) * 0.1
    h = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    c, mean, rstd = gemm.linear_lna(U16, stats, gamma, beta, 1e-5, wq, None, h_out=h, want_stats=True)
    assert float((mean - m_ref).abs().max()) <= 1e-5 and float((rstd / r_ref - 1).abs().max()) <= 1e-5       # exact statistics
    # the normalised rows: the stream enters rounded to bf16 -> |error| <= 2^-9 |u| * rstd * gamma (+ the final rounding)
    bound = (2.0 ** -8 * U.abs() * rstd.unsqueeze(1) * gamma + 2.0 ** -7 * h_ref.float().abs() + 1e-3)
    assert bool(((h.float() - h_ref.float()).abs() <= bound).all())
    want = gemm.linear_tn(h, wq)                      # the same kernel on the rows the fused launch wrote
    assert torch.equal(c, want)
    f = torch.empty(M, 1536, device="cuda", dtype=torch.bfloat16)
    f2, g2, _, _ = gemm.linear_lna(U16, stats, gamma, beta, 1e-5, w1, b1, gelu=True, f_out=f)
    fw, gw = gemm.linear_gelu(h, w1, b1, f_out=torch.empty_like(f))
    assert torch.equal(f2, fw) and torch.equal(g2, gw)


@pytest.mark.parametrize("B,T,nblk,train", [(16, 64, 4, True), (8, 25, 12, True), (32, 64, 3, False)])
def test_stack_with_folded_layernorm_equals_three_kernel_form(B, T, nblk, train):
    from gm3d_amd import gemm, fused, models_mae_learn_loss as M
    torch.manual_seed(B + T)
    stack = M.TransformerEncoder(embed_dim=384, depth=nblk, num_heads=6, drop_path_rate=0.0).cuda()
    norm = torch.nn.LayerNorm(384).cuda()
    stack.train(train)
    x = torch.randn(B, T, 384, device="cuda").bfloat16()
    pos = torch.randn(B, T, 384, device="cuda").bfloat16() * 0.5
    w = torch.randn(B, T, 384, device="cuda")
    res = {}
    was = gemm.FUSE_LN
    try:
        for on in (True, False):
            gemm.FUSE_LN = on
            xs = x.clone().requires_grad_(train)
            with torch.autocast("cuda", dtype=torch.bfloat16), torch.set_grad_enabled(train):
                out = fused.run_stack(stack.blocks, norm, xs, pos, train)
            if train:
                for p in stack.parameters():
                    p.grad = None
                (out.float() * w).sum().backward()
                grads = {n: p.grad.detach().clone() for n, p in stack.named_parameters()}
                grads["x"] = xs.grad.detach().float().clone()
            else:
                grads = {}
            res[on] = (out.detach().float().clone(), grads)
    finally:
        gemm.FUSE_LN = was
    a, b = res[True][0], res[False][0]
    assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max())      # bf16 activations through nblk blocks: one-ulp differences
    assert float((a - b).abs().mean()) <= 2e-3 * float(b.abs().mean())
    for n, gb in res[False][1].items():
        ga = res[True][1][n]
        assert float((ga - gb).abs().max()) <= 4e-2 * float(gb.abs().max()) + 1e-6, n
