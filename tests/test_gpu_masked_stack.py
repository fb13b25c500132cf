"""-m gpu: gm3d_amd/masked_stack.py -- the Point-M2AE block stack as one autograd node, and the visible-first token order.

 * MaskedStackFn against the per-op nodes it replaces (point_m2ae.BlockStack._forward_fused: the same kernels launched one
   autograd node at a time): outputs EQUAL, gradients within the weight-gradient kernels' summation-order band;
 * gm3d_partition_visible / gm3d_select_rows against a torch restatement (stable partition, gather, merge), incl. the overflow flag;
 * the student pass of the whole model in the visible-first order against the in-place order: same reconstruction, scores, losses
   and gradients (fp32: row-local layers are unchanged and the fp32 attention visits the allowed keys in the same order).
The model itself has no reference source (Point-M2AE_SA3D/README.md:1): "parity unpinned", see tests/test_gpu_m2ae.py."""
from contextlib import nullcontext

import pytest
import torch

from tests import clouds

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-12)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("shape", [(4, 256, 192, 0.64), (8, 512, 96, 0.32), (8, 64, 384, 0.0)])
def test_stack_node_equals_per_op_nodes(bf16, shape):
    from gm3d_amd import models_mae_learn_loss as MM
    from gm3d_amd import ops
    from gm3d_amd import point_m2ae as P
    torch.manual_seed(0)
    B, T, C, radius = shape
    stack = P.BlockStack(C, 3, 6, [0.0, 0.05, 0.1]).cuda().train()
    x = torch.randn(B, T, C, device="cuda")
    pos = torch.randn(B, T, C, device="cuda") * 0.3
    cen = clouds.pc_norm(torch.randn(B, T, 3)).cuda()
    vis = torch.rand(B, T, device="cuda") < 0.7
    bits = ops.radius_mask_bits(cen, vis, radius) if radius > 0 else None
    draws = [(torch.rand(B, device="cuda") > 0.2).float() / 0.8 for _ in range(6)]
    from gm3d_amd import masked_stack as S
    res = {}
    was, was_dp, was_g = P.STACK_NODE, MM.drop_path_scale, S.FUSE_GELU
    try:
        for mode in ("node+gelu", "node", "perop"):
            P.STACK_NODE = mode != "perop"
            S.FUSE_GELU = mode == "node+gelu"
            it = iter(draws)
            MM.drop_path_scale = lambda B_, p, training, device: next(it) if p > 0 else None
            xi, pi = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
            for p_ in stack.parameters():
                p_.grad = None
            with (torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else nullcontext()):
                out = stack(xi.bfloat16() if bf16 else xi, pi.bfloat16() if bf16 else pi, bits)
            keep = vis.unsqueeze(-1).float() if bits is not None else torch.ones(B, T, 1, device="cuda")
            (out.float() * keep * torch.linspace(0.5, 1.5, C, device="cuda")).sum().backward()
            res[mode] = (out.detach().float() * keep, xi.grad.clone(), pi.grad.clone(),
                         {k: v.grad.detach().clone() for k, v in stack.named_parameters()})
    finally:
        P.STACK_NODE, MM.drop_path_scale, S.FUSE_GELU = was, was_dp, was_g
    assert torch.equal(res["node"][0], res["perop"][0])          # the same forward kernels in the same order
    # bias + GELU in the fc1 product's epilogue (widths 192 / 384 in bf16): the arithmetic of the two-launch form on the rounded product
    assert torch.equal(res["node+gelu"][0], res["node"][0])
    tol = 2e-2 if bf16 else 2e-5
    for a in ("node", "node+gelu"):
        assert _rel(res[a][1], res["perop"][1]) <= tol and _rel(res[a][2], res["perop"][2]) <= tol
        for k, v in res["perop"][3].items():
            assert _rel(res[a][3][k], v) <= (2e-2 if bf16 else 2e-5), (a, k)


def test_partition_and_select_kernels():
    from gm3d_amd import masked_stack as S
    g = torch.Generator().manual_seed(3)
    flag = S.overflow_flag(torch.device("cuda"))
    for B, T, Tc, p in ((5, 512, 512, 0.5), (7, 256, 96, 0.8), (3, 64, 16, 0.85), (2, 77, 77, 0.3), (4, 300, 128, 0.7)):
        masked = torch.rand(B, T, generator=g) < p
        # keep every cloud within the bound for this part of the test
        for b in range(B):
            v = (~masked[b]).nonzero().flatten()
            if len(v) > Tc:
                masked[b, v[Tc:]] = True
        flag.zero_()
        part = S.partition_visible(masked.cuda(), Tc)
        assert int(flag) == 0
        for b in range(B):
            v = (~masked[b]).nonzero().flatten()
            m = masked[b].nonzero().flatten()
            order = torch.cat([v, m])[:Tc]
            assert torch.equal(part["perm_c"][b].cpu().long(), order)
            want_v = order.clone()
            want_v[len(v):] = -1
            assert torch.equal(part["perm_v"][b].cpu().long(), want_v)
            inv = torch.full((T,), -1, dtype=torch.long)
            inv[v] = torch.arange(len(v))
            assert torch.equal(part["inv_v"][b].cpu().long(), inv)
            assert torch.equal(part["inv_m"][b].cpu().long(), torch.where(masked[b], torch.arange(T), torch.tensor(-1)))
            assert torch.equal(part["vis_c"][b].cpu().bool(), torch.arange(Tc) < len(v))
        for dt, C in ((torch.bfloat16, 96), (torch.float32, 3), (torch.float32, 192), (torch.bfloat16, 3)):
            x = torch.randn(B, T, C, generator=g).to(dt).cuda()
            xc = S.select_rows(x, part["perm_c"])
            assert torch.equal(xc, torch.gather(x, 1, part["perm_c"].long().unsqueeze(-1).expand(-1, -1, C)))
            tok = torch.randn(B, T, C, generator=g).to(dt).cuda()
            merged = S.select_rows(xc, part["inv_v"], tok)
            want = torch.where(masked.cuda().unsqueeze(-1), tok, x)
            assert torch.equal(merged, want)
            zeros = S.select_rows(xc, part["inv_v"])
            assert torch.equal(zeros, torch.where(masked.cuda().unsqueeze(-1), torch.zeros_like(x), x))
    # a bound that is too small is reported
    flag.zero_()
    S.partition_visible(torch.zeros(2, 64, dtype=torch.bool, device="cuda"), 16)
    assert int(flag) == 1
    flag.zero_()


def test_compact_and_merge_gradients():
    from gm3d_amd import masked_stack as S
    torch.manual_seed(1)
    B, T, Tc, C = 4, 256, 96, 192
    masked = torch.rand(B, T, device="cuda") < 0.75
    part = S.partition_visible(masked, Tc)
    x = torch.randn(B, T, C, device="cuda", requires_grad=True)
    tok = torch.randn(B, T, C, device="cuda", requires_grad=True)
    w = torch.randn(B, T, C, device="cuda")
    yc = S.CompactFn.apply(x, part) * 2.0
    out = S.MergeFn.apply(yc, tok, part)
    (out * w).sum().backward()
    vis = (~masked).unsqueeze(-1)
    assert torch.equal(out.detach(), torch.where(vis, 2.0 * x.detach(), tok.detach()))
    assert torch.equal(x.grad, torch.where(vis, 2.0 * w, torch.zeros_like(w)))
    assert torch.equal(tok.grad, torch.where(vis, torch.zeros_like(w), w))


@pytest.mark.parametrize("bf16", [False, True])
def test_visible_first_order_equals_in_place_order(bf16):
    """The student pass of the whole model (multi-scale masks from a 12-of-64 coarse mask) with the stacks in the visible-first
    order (bounds 512 / 96 / 16 rows) against the in-place order."""
    from gm3d_amd import masked_stack as S
    from gm3d_amd import point_m2ae as P
    from gm3d_amd import models_mae_learn_loss as MM
    torch.manual_seed(5)
    B = 4
    model = P.PointM2AE().cuda().train()
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    pts = clouds.gaussian(B, 2048, seed=11).cuda()
    score = torch.rand(B, 64, device="cuda")
    mask, vis_ids, mask_ids = MM.generate_mask_ids(score, mask_ratio=0.8, guide=True, epoch=10, total_epoch=300)
    masked = mask.bool()
    assert vis_ids.shape[1] == 12
    with torch.no_grad():
        group = model.group_divider(pts)
    flag = S.overflow_flag(pts.device)
    flag.zero_()
    res = {}
    was = P.VISIBLE_FIRST
    try:
        for vf in (True, False):
            P.VISIBLE_FIRST = vf
            model.zero_grad(set_to_none=True)
            # BatchNorm running statistics move with every forward: same start for both runs
            sd = {k: v.clone() for k, v in model.state_dict().items()}
            with (torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else nullcontext()):
                out = model(pts, mask=masked, group=group, vis_count=12)
                lo = model.forward_loss(out["rec"], group[0], group[2], out["masks"])
            (lo["Chamfer_mean"] + out["loss_pred"].float().mean()).backward()
            res[vf] = (out["rec"].detach().float(), out["loss_pred"].detach().float(), lo["Chamfer_mean"].detach().float(),
                       {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
            model.load_state_dict(sd)
    finally:
        P.VISIBLE_FIRST = was
    assert int(flag) == 0
    # bf16: 17 blocks of bf16 activations with the attention's partial sums cut at other tile borders: the reconstruction moves by a few
    # bf16 steps of its largest entries, the loss by less
    tol = 3e-2 if bf16 else 2e-5
    assert _rel(res[True][0], res[False][0]) <= (8e-2 if bf16 else tol)
    assert _rel(res[True][1], res[False][1]) <= (8e-2 if bf16 else tol)
    assert _rel(res[True][2], res[False][2]) <= tol
    assert set(res[True][3]) == set(res[False][3])
    gscale = max(float(v.abs().max()) for v in res[False][3].values())
    if bf16:
        # weights in front of a BatchNorm over 16k-1M rows have gradients that are small differences of large sums (tests/test_gpu_m2ae.py):
        # in bf16 they carry tens of percent of noise in EITHER order, so the bar here is on the gradient as a whole
        a = torch.cat([res[True][3][k].flatten().double() for k in sorted(res[False][3])])
        b = torch.cat([res[False][3][k].flatten().double() for k in sorted(res[False][3])])
        assert float(torch.dot(a, b) / (a.norm() * b.norm())) >= 0.98
        assert abs(float(a.norm() / b.norm()) - 1.0) <= 0.05
        return
    for k, v in res[False][3].items():
        # (floor: a bias in front of a BatchNorm has an exactly zero gradient -- both sides hold rounding residue only)
        err = float((res[True][3][k] - v).abs().max()) / max(float(v.abs().max()), 1e-2 * gscale)
        assert err <= 1e-4, (k, err)


def test_back_project_kernel_equals_scatter_form():
    from gm3d_amd import point_m2ae as P
    g = torch.Generator().manual_seed(0)
    B = 5
    idxs = [torch.randint(0, 2048, (B, 512, 16), generator=g), torch.randint(0, 512, (B, 256, 8), generator=g),
            torch.randint(0, 256, (B, 64, 8), generator=g)]
    coarse = torch.rand(B, 64, generator=g) < 0.8
    want = P.back_project(coarse, idxs)                       # CPU: the integer scatter-add form
    got = P.back_project(coarse.cuda(), [i.cuda() for i in idxs])
    assert len(got) == 3
    for a, b in zip(got, want):
        assert a.dtype == torch.bool and torch.equal(a.cpu(), b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_interp3_and_group_max_nodes(dtype):
    from gm3d_amd import heads
    torch.manual_seed(2)
    B, N, S, C1, C2 = 3, 256, 64, 192, 384
    fine = torch.randn(B, N, C1, device="cuda").to(dtype).requires_grad_(True)
    coarse = torch.randn(B, S, C2, device="cuda").to(dtype).requires_grad_(True)
    idx = torch.randint(0, S, (B, N, 3), device="cuda")
    w = torch.rand(B, N, 3, device="cuda")
    w = w / w.sum(-1, keepdim=True)
    out = heads.Interp3Fn.apply(fine, coarse, idx, w)
    g = torch.randn_like(out.float())
    (out.float() * g).sum().backward()
    f64, c64 = fine.detach().double().requires_grad_(True), coarse.detach().double().requires_grad_(True)
    near = torch.gather(c64, 1, idx.reshape(B, N * 3, 1).expand(-1, -1, C2)).view(B, N, 3, C2)
    ref = torch.cat([f64, (near * w.double().unsqueeze(-1)).sum(2)], dim=-1)
    (ref * g.double()).sum().backward()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert _rel(out.detach(), ref.detach()) <= tol
    assert _rel(fine.grad, f64.grad) <= tol and _rel(coarse.grad, c64.grad) <= tol
    # group max: first maximising member, gradient to that member
    x = torch.randn(40, 8, 96, device="cuda").to(dtype)
    x[3, 5] = x[3, 2]                                              # an exact tie: the earlier member wins
    x.requires_grad_(True)
    y = heads.GroupMaxFn.apply(x)
    y.float().sum().backward()
    vals, arg = x.detach().float().max(dim=1)
    assert torch.equal(y.detach().float(), vals)
    first = (x.detach().float() == vals.unsqueeze(1)).float().argmax(dim=1)
    want = torch.zeros_like(x.detach().float()).scatter_(1, first.unsqueeze(1), 1.0)
    assert torch.equal(x.grad.float(), want)


@pytest.mark.parametrize("dtype,C", [(torch.float32, 192), (torch.bfloat16, 96), (torch.float32, 3), (torch.bfloat16, 384)])
def test_where_rows_and_take_rows_nodes(dtype, C):
    from gm3d_amd import heads, ops
    from gm3d_amd.point_m2ae import radius_mask
    torch.manual_seed(4)
    B, T = 5, 77
    masked = torch.rand(B, T, device="cuda") < 0.6
    a = torch.randn(B, T, C, device="cuda").to(dtype).requires_grad_(True)
    alt = torch.randn(B, T, C, device="cuda").to(dtype).requires_grad_(True)
    tok = torch.nn.Parameter(torch.randn(1, 1, C, device="cuda"))
    w = torch.randn(B, T, C, device="cuda")
    m3 = masked.unsqueeze(-1)
    # full alternative
    out = heads.where_rows(masked, a, alt)
    (out.float() * w).sum().backward()
    assert torch.equal(out.detach(), torch.where(m3, alt.detach(), a.detach()))
    assert torch.equal(a.grad.float(), torch.where(m3, torch.zeros_like(w), w).to(dtype).float())
    assert torch.equal(alt.grad.float(), torch.where(m3, w, torch.zeros_like(w)).to(dtype).float())
    # one row for every masked token (the mask token), and zeros
    a.grad = None
    out = heads.where_rows(masked, a, tok)
    (out.float() * w).sum().backward()
    assert torch.equal(out.detach(), torch.where(m3, tok.detach().to(dtype), a.detach()))
    want = torch.where(m3, w.to(dtype).float(), torch.zeros_like(w)).double().sum(dim=(0, 1))
    assert float((tok.grad.double().flatten() - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6
    assert torch.equal(heads.where_rows(masked, a.detach(), None), torch.where(m3, torch.zeros_like(a.detach()), a.detach()))
    # gather by int64 lists with repeats: forward == torch.gather (the backward is csrc/gather.hip's, tested with the model)
    ids = torch.randint(0, T, (B, 200), device="cuda")
    got = heads.take_rows(a.detach(), ids) if C % 8 == 0 else None
    if got is not None:
        assert torch.equal(got, torch.gather(a.detach(), 1, ids.unsqueeze(-1).expand(-1, -1, C)))
    # the radius mask from the "masked" flags directly
    cen = clouds.pc_norm(torch.randn(B, T, 3)).cuda()
    assert torch.equal(ops.radius_mask_bits(cen, None, 0.5, masked=masked), ops.radius_mask_bits(cen, ~masked, 0.5))
