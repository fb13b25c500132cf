"""-m gpu: gm3d_amd/heads.py (pos_embed, pix head, loss-predictor head, mask-token expand, ranking loss) against the
per-op PyTorch modules / expressions of the same package, fp32 at 1e-5 and bf16 against the fp32 result."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def err(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def amax(a):
    return float(a.detach().abs().max())


def test_rank_loss_matches_expression():
    from gm3d_amd import models_mae_learn_loss as M
    m = M.MaskedAutoencoderViT.__new__(M.MaskedAutoencoderViT)
    torch.manual_seed(0)
    p = torch.randn(9, 39, device="cuda", requires_grad=True)
    t = torch.rand(9, 39, device="cuda")
    t[:, 5] = t[:, 6]
    outs = {}
    for fused in (False, True):
        M.FUSED_HEADS = fused
        p.grad = None
        loss = M.MaskedAutoencoderViT.forward_learning_loss(m, p, None, t, relative=True)
        (loss * 3.0).backward()
        outs[fused] = (loss.detach(), p.grad.clone())
    M.FUSED_HEADS = True
    assert err(outs[True][0], outs[False][0]) <= 1e-5 * amax(outs[False][0])
    assert err(outs[True][1], outs[False][1]) <= 2e-5 * amax(outs[False][1])
    nan = M.MaskedAutoencoderViT.forward_learning_loss(m, p, None, torch.ones_like(t), relative=True)
    assert torch.isnan(nan)          # 0/0 like the reference (engine exits on it)


@pytest.mark.parametrize("train", [True, False])
def test_heads_match_modules(train):
    from gm3d_amd import models_mae_learn_loss as M
    torch.manual_seed(3)
    base = M.mae_vit_base_patch16_dec512d8b().cuda()
    with torch.no_grad():
        bn = base.increase_dim_2[1]
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3); bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
        base.mask_token.normal_()
    B, L = 6, 64
    center = torch.rand(B, L, 3, device="cuda") - 0.5
    xr = torch.randn(B, L, 384, device="cuda")
    w_pos, w_pix, w_lp = (torch.randn(B, L, 384, device="cuda"), torch.randn(B, L, 96, device="cuda"), torch.randn(B, L, device="cuda"))

    def run(fused, bf16):
        m = copy.deepcopy(base).train(train)
        M.FUSED_HEADS = fused
        x = xr.clone().requires_grad_(train)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16), torch.set_grad_enabled(train):
            pos = m.embed_pos(center)
            if fused:
                from gm3d_amd import heads
                pix = heads.LinearBiasFn.apply(x, m.increase_dim_just_network_without_feature[0].weight,
                                               m.increase_dim_just_network_without_feature[0].bias, heads._adt())
                tok = heads.ExpandRowsFn.apply(m.mask_token, B, 5, pos.dtype)
            else:
                c = m.increase_dim_just_network_without_feature[0]
                pix = torch.nn.functional.linear(x, c.weight.squeeze(-1), c.bias)
                tok = m.mask_token.expand(B, 5, -1).to(pos.dtype)
            lp = m._loss_pred_head(x)
            if train:
                ((pos.float() * w_pos).sum() + (pix.float() * w_pix).sum() + (lp.float() * w_lp).sum()
                 + (tok.float() * w_pos[:, :5]).sum()).backward()
        M.FUSED_HEADS = True
        grads = {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}
        if train:
            grads["x"] = x.grad.detach().double()
        bufs = {k: b.detach().double() for k, b in m.increase_dim_2.named_buffers()}
        return {"pos": pos.detach().double(), "pix": pix.detach().double(), "lp": lp.detach().double()}, grads, bufs

    ro, rg, rb = run(False, False)
    fo, fg, fb = run(True, False)
    for k in ro:
        assert err(fo[k], ro[k]) <= 1e-5 * amax(ro[k]), k
    gn = sum(float(v.pow(2).sum()) for v in rg.values()) ** 0.5 if rg else 0.0
    assert set(fg) == set(rg)
    for k in rg:
        assert err(fg[k], rg[k]) <= 3e-5 * amax(rg[k]) + 1e-6 * gn, k
    for k in rb:
        assert err(fb[k], rb[k]) <= 1e-5 * amax(rb[k]) + 1e-7, k
    mo, mg, _ = run(False, True)
    bo, bg, _ = run(True, True)
    for k in ro:
        assert err(bo[k], ro[k]) <= 2 * err(mo[k], ro[k]) + 2e-2 * amax(ro[k]), k
    for k in rg:
        assert err(bg[k], rg[k]) <= 2 * err(mg[k], rg[k]) + 2e-2 * amax(rg[k]) + 1e-6 * gn, k


@pytest.mark.parametrize("B,L,ratio,epoch", [(128, 64, 0.6, 200), (5, 64, 0.6, 0), (7, 64, 0.75, 399), (3, 40, 0.5, 100), (2, 64, 0.6, 200)])
def test_mask_select_kernel(B, L, ratio, epoch):
    """gm3d_mask_select against the argsort/scatter formulation of generate_mask (P/:744-784) and split_ids, including
    ties in loss_pred and in the noise (resolved towards the lower index on both sides via stable sorts)."""
    from gm3d_amd import models_mae_learn_loss as M
    m = M.mae_vit_base_patch16_dec512d8b()
    g = torch.Generator().manual_seed(B * 1000 + L)
    lp = torch.randn(B, L, generator=g)
    noise = torch.rand(B, L, generator=g)
    if B == 2:                       # heavy ties
        lp = (lp * 2).round() / 2
        noise = (noise * 8).floor() / 8
    len_keep = int(L * (1 - ratio))
    len_loss = int((L - len_keep) * (float((epoch + 1) / 400) * 0.5))
    nz = noise.clone()
    if len_loss > 0:
        forced = torch.argsort(lp, dim=1, stable=True)[:, L - len_loss:]
        nz.scatter_(1, forced, float("inf"))
    keep = torch.argsort(nz, dim=1, stable=True)[:, :len_keep]
    want = torch.ones(B, L)
    want.scatter_(1, keep, 0.0)
    mask, vis, msk = m.generate_mask_ids(lp.cuda(), ratio, True, epoch, 400, noise.cuda())
    assert torch.equal(mask.cpu(), want)
    wv, wm = M.split_ids(want.bool(), len_keep)
    assert torch.equal(vis.cpu(), wv) and torch.equal(msk.cpu(), wm)
    assert torch.equal(m.generate_mask(lp.cuda(), ratio, epoch=epoch, total_epoch=400, noise=noise).cpu(), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [25, 64, 1])
def test_token_assemble(dtype, V):
    """TokenAssembleFn == take/cat of the module path, forward and backward (exact: pure data movement + one add)."""
    from gm3d_amd import heads
    from gm3d_amd import models_mae_learn_loss as M
    B, L, C = 7, 64, 384
    g = torch.Generator().manual_seed(V)
    mask = torch.zeros(B, L, dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(L, generator=g)[: L - V]] = True
    vis_ids, mask_ids = M.split_ids(mask.cuda(), V)
    tok = torch.randn(B, L, C, generator=g).to(dtype).cuda().requires_grad_(True)
    pos = torch.randn(B, L, C, generator=g).to(dtype).cuda().requires_grad_(True)
    w = [torch.randn(B, V, C, generator=g).to(dtype).cuda(), torch.randn(B, V, C, generator=g).to(dtype).cuda(),
         torch.randn(B, L, C, generator=g).to(dtype).cuda()]
    xv, pv, pf = heads.token_assemble(tok, pos, vis_ids, mask_ids)
    (xv * w[0]).sum().backward(retain_graph=True); (pv * w[1]).sum().backward(retain_graph=True); (pf * w[2]).sum().backward()
    got = (xv.detach().clone(), pv.detach().clone(), pf.detach().clone(), tok.grad.clone(), pos.grad.clone())
    tok.grad = pos.grad = None
    rxv, rpv = M.take(tok, vis_ids), M.take(pos, vis_ids)
    rpf = torch.cat([M.take(pos, vis_ids), M.take(pos, mask_ids)], dim=1)
    ((rxv * w[0]).sum() + (rpv.float() * w[1].float()).sum() + (rpf.float() * w[2].float()).sum()).backward()
    assert torch.equal(got[0], rxv) and torch.equal(got[1], rpv) and torch.equal(got[2], rpf)
    assert torch.equal(got[3], tok.grad)
    tol = 0.0 if dtype == torch.float32 else 2.0 ** -7 * float(pos.grad.abs().max())     # bf16: the module path sums in another order
    assert float((got[4].float() - pos.grad.float()).abs().max()) <= tol + 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,M", [(4, 39), (128, 39), (3, 1), (2, 64)])
def test_patch_chamfer_loss_equals_the_op_chain(dtype, B, M):
    """heads.PatchChamferLossFn (gather + cast + Chamfer + both means in one pass, csrc/chamfer.hip) against the op chain of
    forward_loss (models_mae_learn_loss.py:384-412 restated with the separate kernels): per-patch losses and the gradient agree to
    fp32 summation order (the distances and argmins are the same arithmetic), also on a batch-strided view of the head's output."""
    from gm3d_amd import heads, ops
    from gm3d_amd import models_mae_learn_loss as M_
    L = 64
    g = torch.Generator().manual_seed(B * 7 + M)
    pix = (torch.randn(B, L, 96, generator=g) * 0.3).cuda().to(dtype).requires_grad_(True)
    target = (torch.randn(B, L, 32, 3, generator=g) * 0.3).cuda()
    order = torch.stack([torch.randperm(L, generator=g) for _ in range(B)]).cuda()
    mask_ids = order[:, L - M:]
    pred = pix[:, L - M:]
    assert heads.patch_chamfer_loss_supported(pred, target, mask_ids)
    mean, matrix = heads.PatchChamferLossFn.apply(pred, target, mask_ids)
    (mean * 3.0).backward()
    got_g = pix.grad.clone()
    pix.grad = None
    tg = M_.take(target, mask_ids).reshape(-1, 32, 3)
    loss = ops.ChamferDistanceL2()(pix[:, L - M:].reshape(-1, 32, 3).to(torch.float32), tg).reshape(B, -1, 32)
    want_matrix = loss.mean(dim=-1)
    want_mean = want_matrix.mean()
    (want_mean * 3.0).backward()
    assert float((matrix - want_matrix).abs().max()) <= 2e-6 * float(want_matrix.abs().max())
    assert abs(float(mean) - float(want_mean)) <= 2e-6 * abs(float(want_mean))
    scale = float(pix.grad.float().abs().max())
    tol = 2e-6 if dtype == torch.float32 else 2.0 ** -8
    assert float((got_g.float() - pix.grad.float()).abs().max()) <= tol * scale
    assert float(got_g[:, :L - M].abs().max()) == 0.0 if M < L else True


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_take_rows_deterministic_backward(dtype):
    """heads.take_rows: gather with repeated indices; the backward (csrc/gather.hip: inverse lists in ascending j, then an ordered sum
    per source row) equals an fp64 index_add of the same gradients within one rounding and is BIT-IDENTICAL from run to run (PyTorch's
    scatter-add with colliding atomics is not), including rows nobody reads (zeros)."""
    from gm3d_amd import heads
    g = torch.Generator(device="cuda").manual_seed(7)
    for B, S, J, C in ((5, 64, 768, 384), (3, 512, 2048, 96), (2, 256, 512, 192), (1, 7, 40, 8)):
        x = torch.randn(B, S, C, device="cuda", generator=g).to(dtype).requires_grad_(True)
        ids = torch.randint(0, S - 1, (B, J), device="cuda", generator=g)            # row S-1 is never read
        dy = torch.randn(B, J, C, device="cuda", generator=g).to(dtype)
        y = heads.take_rows(x, ids)
        assert torch.equal(y, torch.gather(x, 1, ids.unsqueeze(-1).expand(-1, -1, C)))
        y.backward(dy)
        first = x.grad.clone()
        ref = torch.zeros(B, S, C, device="cuda", dtype=torch.float64)
        ref.scatter_add_(1, ids.unsqueeze(-1).expand(-1, -1, C), dy.double())
        tol = 1e-6 if dtype == torch.float32 else 1e-2
        assert float((first.double() - ref).abs().max()) <= tol * max(float(ref.abs().max()), 1.0)
        assert bool((first[:, S - 1] == 0).all())
        for _ in range(3):
            x.grad = None
            heads.take_rows(x, ids).backward(dy)
            assert torch.equal(x.grad, first)


def test_rank_loss_tail_folds_slice_and_reductions():
    """heads.rank_loss_tail(full, M, target): the ranking loss on full[:, -M:] with the slice, the .sum(0), the division and (backward)
    the scaling + zero-filled slice gradient inside our launches == heads.rank_loss on the sliced tensor (loss within fp32 summation
    order of the 128 per-sample terms; gradient identical up to that scale), zeros in front."""
    from gm3d_amd import heads
    g = torch.Generator(device="cuda").manual_seed(3)
    for B, L, M in ((128, 64, 39), (7, 64, 64), (33, 40, 2)):
        full = torch.randn(B, L, device="cuda", generator=g).requires_grad_(True)
        target = torch.rand(B, M, device="cuda", generator=g)
        a = heads.rank_loss_tail(full, M, target)
        (a * 1.7).backward()
        ga = full.grad.clone()
        full.grad = None
        b = heads.rank_loss(full[:, -M:], target)
        (b * 1.7).backward()
        assert abs(float(a) - float(b)) <= 2e-6 * abs(float(b)) + 1e-7
        assert bool((ga[:, :L - M] == 0).all())
        assert float((ga - full.grad).abs().max()) <= 2e-6 * float(full.grad.abs().max()) + 1e-12


def test_drop_path_scales_kernel_equals_torch_ops():
    """gm3d_drop_path_scales == (u + keep).floor_().div_(keep) bit for bit (the same three IEEE operations)."""
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    g = torch.Generator(device="cuda").manual_seed(5)
    u = torch.rand(34, 128, device="cuda", generator=g)
    keep = (1.0 - torch.linspace(0.003, 0.1, 34, device="cuda")).unsqueeze(1).contiguous()
    out = torch.empty_like(u)
    check(lib.gm3d_drop_path_scales(_ptr(u), _ptr(keep), 34, 128, _ptr(out), _stream()), "dp")
    assert torch.equal(out, (u + keep).floor_().div_(keep))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pad_cols(dtype):
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream, _DT
    src = torch.randn(1000, 104, device="cuda").to(dtype)[:, :96]
    dst = torch.full((1000, 128), 7.0, device="cuda", dtype=dtype)
    check(lib.gm3d_pad_cols(_ptr(src), src.stride(0), 1000, 96, _ptr(dst), 128, _DT[dtype], _stream()), "pad")
    assert torch.equal(dst[:, :96], src) and bool((dst[:, 96:] == 0).all())


def test_patch_loss_full_prediction_gradient():
    """PatchChamferLossFn(full=True): the loss of pix_pred[:, -M:] taken from the whole (B,L,96) prediction; backward = the gradient of
    the whole tensor, zeros for the visible patches, identical to the sliced form + autograd's slice backward."""
    from gm3d_amd import heads
    g = torch.Generator(device="cuda").manual_seed(9)
    B, L, M = 16, 64, 39
    for dtype in (torch.bfloat16, torch.float32):
        full = (torch.randn(B, L, 96, device="cuda", generator=g) * 0.3).to(dtype).requires_grad_(True)
        target = torch.randn(B, L, 32, 3, device="cuda", generator=g) * 0.3
        ids = torch.stack([torch.randperm(L, device="cuda", generator=g)[:M].sort().values for _ in range(B)])
        m1, mat1 = heads.PatchChamferLossFn.apply(full, target, ids, True)
        (m1 * 2.0).backward()
        g1 = full.grad.clone()
        full.grad = None
        m2, mat2 = heads.PatchChamferLossFn.apply(full[:, -M:], target, ids)
        (m2 * 2.0).backward()
        assert torch.equal(m1, m2) and torch.equal(mat1, mat2)
        assert torch.equal(g1, full.grad) and bool((g1[:, :L - M] == 0).all())
