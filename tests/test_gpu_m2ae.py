"""-m gpu: Point-M2AE + GeoMask3D (BASELINE config #4) -- gm3d_amd/point_m2ae.py on the GPU against the functional
CPU restatement oracle/hier_ref.py with identical name-derived weights, fp32, B=2, N=2048: teacher scores, guided mask, multi-scale
masks, reconstruction, predicted losses, both losses and parameter gradients <= 2e-5 relative.  "Parity unpinned": the reference
has no source for this model (Point-M2AE_SA3D/README.md:1); the oracle is our own restatement of the configuration + paper."""
from types import SimpleNamespace

import pytest
import torch

from tests import clouds

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_radius_mask_bits_kernel():
    from gm3d_amd import ops
    from gm3d_amd.point_m2ae import radius_mask
    from oracle import hier_ref as HR
    g = torch.Generator().manual_seed(0)
    for G, radius in ((512, 0.32), (256, 0.64), (64, 1.28), (77, 0.5)):
        c = clouds.pc_norm(torch.randn(3, G, 3, generator=g))
        vis = torch.rand(3, G, generator=g) < 0.6
        want = ~(vis[:, :, None] & vis[:, None, :]) | HR.far_mask(c, radius)
        assert torch.equal(radius_mask(c.cuda(), radius).cpu(), HR.far_mask(c, radius))
        got = ops.radius_mask_bits(c.cuda(), vis.cuda(), radius)
        assert torch.equal(got, ops.pack_mask(want.cuda()))
        assert torch.equal(ops.radius_mask_bits(c.cuda(), None, radius), ops.pack_mask(HR.far_mask(c, radius).cuda()))


# Max-pool winners (including EXACT ties, which do occur: torch.amax shares the gradient between tied maxima, a plain argmax does
# not), ReLU / LeakyReLU sign patterns (a BatchNorm bias gradient in the loss head is a sum over only B*64 rows: ONE
# pre-activation within 1e-5 of zero moves it by 1 %), Chamfer nearest neighbours, the guided mask and the sign pattern of the
# ranking loss are DISCRETE decisions: a
# near-tie that fp32 on the GPU resolves differently from fp32 / fp64 on the CPU puts the two runs on different smooth branches, and
# their gradients then differ by O(1e-3) in a few rows although every kernel is right (about half of all seeds: round 2 hand-picked
# seeds without such a flip).  Now the oracle TAKES the product's decisions (oracle/hier_ref.py `decisions=`; the product exposes
# them: point_m2ae.POOL_TAPS, its reconstruction, its per-token target), so every seed compares the same branch and a wrong backward
# kernel cannot hide behind "a decision flipped".  The decision logic itself is checked separately: guided mask recomputed by the
# oracle from the product's scores, argmax / argmin / grouping kernels bit-exact in their own tests.  12 CONSECUTIVE seeds.
def _compare_with_oracle(seed):
    """-> rows (name, e_prod, e_cpu, zero_gradient) for every parameter with a gradient, after the forward checks."""
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import point_m2ae as P
    from oracle import hier_ref as HR
    from oracle import model_ref as R
    from oracle import ops as oracle_ops
    oracle_ops.build()
    epoch = 0 if seed % 2 == 0 else 200          # both branches of the guided mask (len_loss = 0 / > 0)
    B, total = 2, 300
    pts = clouds.gaussian(B, 2048, seed=seed)
    noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(5))
    model = P.PointM2AE()
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    R.det_fill_(model, seed=3)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    teacher_sd = {k: v.clone() for k, v in R.det_fill_(P.PointM2AE(), seed=4).state_dict().items()}
    model = model.cuda().train()
    ema = E.ModelEma(model, 0.999)
    ema.ema.load_state_dict(teacher_sd)
    P.POOL_TAPS, P.ACT_TAPS = [], []
    try:
        out = P.pretrain_forward(model, ema.ema, pts.cuda(), epoch, total, mask_noise=noise.cuda())
        pool_idx, act_signs = [t.cpu() for t in P.POOL_TAPS], [t.cpu() for t in P.ACT_TAPS]
    finally:
        P.POOL_TAPS = P.ACT_TAPS = None
    assert len(pool_idx) == 6 and len(act_signs) == 9
    out["loss"].backward()
    # the product's decisions
    mask = out["mask"].cpu()
    G1, k1 = out["rec"].shape[1], out["rec"].shape[2]
    group = HR.hierarchical_group(pts)
    rec_cpu = out["rec"].detach().float().cpu().reshape(B * G1, k1, 3)
    _, _, i1, i2 = oracle_ops.chamfer(rec_cpu, group[0][1].reshape(B * G1, k1, 3).float())    # == the product's kernel, bit for bit
    decisions = {"pool_idx": pool_idx, "act_signs": act_signs, "nn_idx": (i1.long(), i2.long()),
                 "rank_target": out["matrix"].detach().float().cpu()}
    del HR.DECISION_MARGINS[:]
    ref = HR.m2ae_pretrain_forward(sd, teacher_sd, pts, epoch, total, noise, mask=mask, decisions=decisions)
    ref["loss"].backward()
    # the injected decisions are the oracle's own except at near-ties: a sign taken over against a pre-activation, or a
    # pool winner below the maximum, by more than fp32 rounding of the tensor's scale would be a WRONG decision of the product
    worst = {}
    for kind, m in HR.DECISION_MARGINS:
        worst[kind] = max(worst.get(kind, 0.0), m)
    assert len(HR.DECISION_MARGINS) >= 15 and worst.get("sign", 0.0) <= 1e-4 and worst.get("pool", 0.0) <= 1e-4, worst
    # nearest neighbours: the oracle's own choice on ITS reconstruction differs from the injected one only at near-ties
    _, _, i1o, i2o = oracle_ops.chamfer(ref["rec"].detach().float().reshape(B * G1, k1, 3), group[0][1].reshape(B * G1, k1, 3).float())
    assert float((i1o.long() != i1.long()).float().mean()) <= 1e-3 and float((i2o.long() != i2.long()).float().mean()) <= 1e-3
    # decision logic: the teacher's scores agree, and the oracle's mask rule applied to the PRODUCT's scores gives the product's mask
    assert _rel(out["teacher_loss_pred"], ref["teacher_loss_pred"]) <= 2e-5
    want_mask = HR.guided_mask(out["teacher_loss_pred"].detach().float().cpu(), noise, HR.CFG["mask_ratio"], epoch, total)
    assert torch.equal(mask, want_mask) and int(mask[0].sum()) == 52
    assert _rel(out["matrix"], ref["matrix"]) <= 2e-5
    for k in ("loss_chfr", "loss_learn", "loss"):
        assert _rel(out[k], ref[k]) <= 2e-5, (k, float(out[k]), float(ref[k]))
    # Gradients.  Several are ill-conditioned in fp32 (weights in front of a BatchNorm over 16k-1M rows: a small difference of large
    # sums; biases there have an exactly zero gradient), so "within 5e-5" is not a meaningful bar for them on EITHER side.  The
    # same computation in fp64 -- same decisions -- is the truth; the product must be as close to it as the fp32 CPU restatement
    # is (x3), or 5e-5.
    sd64 = {k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point else v.detach()) for k, v in sd.items()}
    t64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in teacher_sd.items()}
    ref64 = HR.m2ae_pretrain_forward(sd64, t64, pts, epoch, total, noise, mask=mask, decisions=decisions)
    ref64["loss"].backward()
    gscale = max(float(t.grad.abs().max()) for t in sd64.values() if torch.is_tensor(t) and t.grad is not None)
    rows = []
    for name, p in model.named_parameters():
        g64, g32 = sd64[name].grad, sd[name].grad
        if p.grad is None:
            assert g64 is None or float(g64.abs().max()) <= 1e-9 * gscale, name
            continue
        scale = max(float(g64.abs().max()), 1e-4 * gscale)
        e_prod = float((p.grad.detach().cpu().double() - g64).abs().max()) / scale
        e_cpu = float((g32.double() - g64).abs().max()) / scale
        rows.append((name, e_prod, e_cpu, float(g64.abs().max()) <= 1e-9 * gscale))
    return rows


@pytest.mark.parametrize("seed", list(range(20, 32)))
def test_m2ae_forward_backward_against_oracle(seed):
    rows = _compare_with_oracle(seed)
    bad = {}
    for name, e_prod, e_cpu, zero in rows:
        if zero:
            # an exactly zero gradient (a bias in front of a BatchNorm): what either side reports is the rounding residue of a
            # sum of cancelling terms, whose size depends on the summation order only -- bar: 3e-7 of the largest gradient
            if e_prod > max(3e-3, 3.0 * e_cpu):
                bad[name] = (e_prod, e_cpu)
        elif e_prod > max(5e-5, 3.0 * e_cpu):
            bad[name] = (e_prod, e_cpu)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]
    assert len(rows) > 150


def test_m2ae_bf16_step_runs_and_learns():
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import point_m2ae as P
    torch.manual_seed(0)
    model = P.PointM2AE().cuda().train()
    ema = E.ModelEma(model, 0.999)
    opt = E.build_optimizer(model, lr=5e-4, flat=True, model_ema=ema)
    args = SimpleNamespace(bf16=True, epochs=300)
    pts = clouds.gaussian(8, 2048, seed=1).cuda()
    before = model.rec_head.weight.detach().clone()
    losses = []
    for i in range(6):
        o = P.pretrain_step(model, ema, opt, pts.clone(), 10, args, augment=False)
        losses.append(float(o["loss_chfr"]))
        assert all(float(o[k]) == float(o[k]) for k in ("loss", "loss_learn", "grad_norm"))
    assert not torch.equal(model.rec_head.weight, before)
    assert losses[-1] < losses[0]


def test_m2ae_step_replays_like_eager_at_full_batch():
    """The whole Point-M2AE step (clip + AdamW + EMA included) captured as a hipGraph and replayed on fresh inputs at the bench's
    B = 128 -- the size at which PyTorch's own bias-gradient reductions came back non-finite from the second replay on
    (tools/m2ae_step_diag.py; the model's Linear layers now use our column-sum kernel).  A twin model stepping eagerly on the same
    inputs must see the SAME losses, weights and teacher, to the bit: every reduction of the step has a fixed order (the repeated-index
    gathers take their backward from csrc/gather.hip, not from PyTorch's colliding atomics), so replay and eager execution are the same
    arithmetic."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import point_m2ae as P
    B = 128
    args = SimpleNamespace(bf16=True, epochs=300)
    pool = [clouds.gaussian(B, 2048, seed=900 + i).cuda() for i in range(5)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(40 + i)).cuda() for i in range(5)]
    twins = []
    for _ in range(2):
        torch.manual_seed(11)
        m = P.PointM2AE().cuda().train()
        for mod in m.modules():
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        twins.append((m, ema, E.build_optimizer(m, lr=1e-3, flat=True, model_ema=ema)))
    (ma, ea, oa), (mb, eb, ob) = twins
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):                      # the eager iterations before a capture: optimizer steps on both twins
            P.pretrain_step(ma, ea, oa, pool[i].clone(), 100, args, mask_noise=noise[i], augment=False)
            P.pretrain_step(mb, eb, ob, pool[i].clone(), 100, args, mask_noise=noise[i], augment=False)
        torch.cuda.synchronize()
        static_in, static_noise = pool[0].clone(), noise[0].clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = P.pretrain_step(mb, eb, ob, static_in, 100, args, mask_noise=static_noise, augment=False)
    torch.cuda.current_stream().wait_stream(side)
    # what the allocator hands out after the capture: sentinels (a replay must not write there) and NaN blocks (nor read there)
    guards = [torch.full((n,), 12345.0, device="cuda") for n in (1, 2, 8, 64, 1024, 1 << 18) for _ in range(64)]
    ints = [torch.full((), 777, dtype=torch.int64, device="cuda") for _ in range(128)]
    poison = [torch.full((n,), float("nan"), device="cuda") for n in (1, 4, 16, 96, 384, 1536, 1 << 14, 1 << 20) for _ in range(64)]
    for i in range(2, 5):
        want = P.pretrain_step(ma, ea, oa, pool[i].clone(), 100, args, mask_noise=noise[i], augment=False)
        static_in.copy_(pool[i])
        static_noise.copy_(noise[i])
        g.replay()
        torch.cuda.synchronize()
        for k in ("loss_chfr", "loss_learn", "grad_norm"):
            a, b = float(want[k]), float(out[k])
            assert b == b and abs(b) != float("inf"), (i, k, b)
            assert a == b, (i, k, a, b)
        assert bool(torch.isfinite(ob.P).all()) and bool(torch.isfinite(ob.E).all())
    assert torch.equal(oa.P, ob.P) and torch.equal(oa.E, ob.E)
    assert all(bool((g == 12345.0).all()) for g in guards) and all(int(t) == 777 for t in ints)
    del poison
    assert int(out["vis_overflow"]) == 0          # the static bounds of the visible-first order held on every replayed batch


def test_m2ae_bf16_mode_tracks_fp32_mode():
    """Throughput mode (bf16 autocast: fused level-0 embed, bf16 weight shadows, LayerNorm / Linear outputs in bf16) against parity
    mode (fp32, the path pinned to the oracle above) on the same weights, clouds and mask noise: the teacher's scores rank the
    tokens alike and both losses agree within the bf16 band."""
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import point_m2ae as P
    torch.manual_seed(21)
    model = P.PointM2AE().cuda().train()
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    ema = E.ModelEma(model, 0.999)
    pts = clouds.gaussian(8, 2048, seed=77).cuda()
    noise = torch.rand(8, 64, generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        ref = P.pretrain_forward(model, ema.ema, pts, 100, 300, mask_noise=noise)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = P.pretrain_forward(model, ema.ema, pts, 100, 300, mask_noise=noise)
    a, b = ref["teacher_loss_pred"].float(), got["teacher_loss_pred"].float()
    assert float((a - b).abs().max()) <= 3e-2 * float(a.abs().max())
    # the guided mask keeps the highest-scoring tokens: where the two modes disagree the scores must be near ties
    same = (ref["mask"] == got["mask"]).float().mean()
    assert float(same) >= 0.9
    if bool((ref["mask"] == got["mask"]).all()):
        for k in ("loss_chfr", "loss_learn"):
            assert abs(float(ref[k]) - float(got[k])) <= 3e-2 * abs(float(ref[k])), k


@pytest.mark.parametrize("bf16", [False, True])
def test_fused_block_stack_equals_per_op_blocks(bf16):
    """point_m2ae.BlockStack with the residual sums inside the LayerNorm kernels (FUSED_BLOCKS) against the per-op blocks on the same
    weights, DropPath factors injected: outputs and every parameter / input gradient.  fp32: 2e-5; bf16: the fused path rounds a
    residual sum once where the per-op path rounds twice (2e-2 of the tensor scale)."""
    from contextlib import nullcontext
    from gm3d_amd import models_mae_learn_loss as MM
    from gm3d_amd import ops
    from gm3d_amd import point_m2ae as P
    torch.manual_seed(0)
    B, T, C = 4, 256, 192
    stack = P.BlockStack(C, 3, 6, [0.0, 0.05, 0.1]).cuda().train()
    x = torch.randn(B, T, C, device="cuda")
    pos = torch.randn(B, T, C, device="cuda") * 0.3
    cen = clouds.pc_norm(torch.randn(B, T, 3)).cuda()
    vis = torch.rand(B, T, device="cuda") < 0.7
    bits = ops.radius_mask_bits(cen, vis, 0.64)
    draws = [(torch.rand(B, device="cuda") > 0.2).float() / 0.8 for _ in range(6)]
    res = {}
    was, was_dp, was_dpm = P.FUSED_BLOCKS, MM.drop_path_scale, MM.drop_path
    try:
        for fused in (True, False):
            P.FUSED_BLOCKS = fused
            it = iter(draws)
            MM.drop_path_scale = lambda B_, p, training, device: next(it) if p > 0 else None
            MM.drop_path = lambda t, p, training: t * next(it).view(-1, 1, 1).to(t.dtype) if p > 0 else t      # the per-op DropPath module
            xi, pi = x.clone().requires_grad_(True), pos.clone().requires_grad_(True)
            for p_ in stack.parameters():
                p_.grad = None
            with (torch.autocast("cuda", dtype=torch.bfloat16) if bf16 else nullcontext()):
                out = stack(xi.bfloat16() if bf16 else xi, pi.bfloat16() if bf16 else pi, bits)
            keep = vis.unsqueeze(-1).float()                   # rows of invisible tokens are garbage by design: not compared
            (out.float() * keep * torch.linspace(0.5, 1.5, C, device="cuda")).sum().backward()
            res[fused] = (out.detach().float() * keep, xi.grad.clone(), pi.grad.clone(),
                          {k: v.grad.detach().clone() for k, v in stack.named_parameters()})
    finally:
        P.FUSED_BLOCKS, MM.drop_path_scale, MM.drop_path = was, was_dp, was_dpm
    tol = 2e-2 if bf16 else 2e-5
    rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
    assert rel(res[True][0], res[False][0]) <= tol
    assert rel(res[True][1], res[False][1]) <= 2 * tol and rel(res[True][2], res[False][2]) <= 2 * tol
    for k, v in res[False][3].items():
        assert rel(res[True][3][k], v) <= (4e-2 if bf16 else 5e-5), k


def test_graphed_step_with_staged_grouping_equals_eager():
    """point_m2ae.GraphedM2AEStep: the training graph + the next batch's grouping graph on a second stream, fed with look-ahead
    (next_pts) and without -- a twin stepping eagerly on the same inputs sees EQUAL losses, weights and teacher (augmentation and
    DropPath off, mask noise injected: the random draws of the two orders are not the same stream)."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import point_m2ae as P
    B = 16
    args = SimpleNamespace(bf16=True, epochs=300)
    pool = [clouds.gaussian(B, 2048, seed=700 + i).cuda() for i in range(6)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(60 + i)).cuda() for i in range(6)]
    twins = []
    for _ in range(2):
        torch.manual_seed(13)
        m = P.PointM2AE().cuda().train()
        for mod in m.modules():
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        twins.append((m, ema, E.build_optimizer(m, lr=1e-3, flat=True, model_ema=ema)))
    (ma, ea, oa), (mb, eb, ob) = twins
    for _ in range(2):          # the eager iterations before a capture, on both twins with the same inputs
        P.pretrain_step(ma, ea, oa, pool[0].clone(), 100, args, mask_noise=noise[0], augment=False)
        P.pretrain_step(mb, eb, ob, pool[0].clone(), 100, args, mask_noise=noise[0], augment=False)
    g = P.GraphedM2AEStep(mb, eb, ob, args, pool[0], 100, augment=False, inject_mask_noise=True, warmup_iters=0)
    for i in range(1, 6):
        want = P.pretrain_step(ma, ea, oa, pool[i].clone(), 100, args, mask_noise=noise[i], augment=False)
        nxt = pool[i + 1] if (i + 1 < 6 and i % 2 == 1) else None          # with and without look-ahead
        got = g(pool[i], mask_noise=noise[i], next_pts=nxt)
        torch.cuda.synchronize()
        for k in ("loss_chfr", "loss_learn", "grad_norm"):
            assert float(want[k]) == float(got[k]), (i, k, float(want[k]), float(got[k]))
    assert torch.equal(oa.P, ob.P) and torch.equal(oa.E, ob.E)
