"""-m gpu: the captured hipGraph step must replay to the same trajectory as eager execution.  Randomness is removed
(DropPath off, augmentation off, mask noise injected) so the two runs are comparable step by step.  This is the guard
against the stack's replay-unsafe reductions (engine_pretrain.GraphedPretrainStep docstring)."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf16", [True, False])
def test_graph_replay_equals_eager(bf16):
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    B, steps = 16, 6
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=bf16, accum_iter=1, lr=2e-4, min_lr=0.0,
                           warmup_epochs=40)
    pool = [clouds.uniform(B, 1024, 50 + i).cuda() for i in range(3)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(i)).cuda() for i in range(steps + 4)]

    def build():
        torch.manual_seed(0)
        m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
        for mod in m.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        # bf16 case: the production configuration (FlatAdamWEma); fp32 case: torch.optim.AdamW (capturable)
        opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema) if bf16 else E.build_optimizer(m, lr=2e-4, capturable=True)
        return m, ema, opt

    m, ema, opt = build()
    eager = []
    for i in range(steps + 4):
        o = E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, args, mask_noise=noise[i], augment=False)
        eager.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])

    m, ema, opt = build()
    # capture runs 3 eager warm-up steps + nothing during capture: feed them the same first inputs as the eager run
    g = E.GraphedPretrainStep.__new__(E.GraphedPretrainStep)
    g.static_in = pool[0].clone()
    g.static_noise = noise[0].clone()
    g.graph2 = g.grad_sync = None
    for i in range(3):
        E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, args, mask_noise=noise[i], augment=False)
    torch.cuda.synchronize()
    g.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g.graph):
        g.out = E.pretrain_step(m, ema, opt, g.static_in, 200, args, mask_noise=g.static_noise, augment=False)
    got = []
    for i in range(3, steps + 4):
        o = g(pool[i % 3], noise[i])
        torch.cuda.synchronize()
        got.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
    # identical kernels on identical operands, every reduction in a fixed order (the index gathers scatter over permutations:
    # their backward atomics never collide) -> the trajectories are EQUAL, not close (tools/eager_vs_graph_diag.py)
    assert all(x == x for a in got for x in a)
    assert got == eager[3:], (got, eager[3:])


def test_two_graph_data_parallel_path_on_one_gpu():
    """The data-parallel capture (forward+backward graph | eager all-reduce of the flat buckets | update graph) with a
    world of one: gradients accumulate into GradSync's flat views inside the first graph; results must equal eager."""
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    B = 8
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0,
                           warmup_epochs=40)
    pool = [clouds.uniform(B, 1024, 70 + i).cuda() for i in range(3)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(i)).cuda() for i in range(8)]
    res = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(0)
        m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
        for mod in m.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema)
        sync = E.GradSync.from_flat(opt, bucket_bytes=32 << 20)     # buckets = chunks of the optimizer's flat gradient buffer
        assert len(sync.buckets) == 5
        hist = []
        if mode == "eager":
            for i in range(8):
                o = E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, args, grad_sync=sync, mask_noise=noise[i], augment=False)
                hist.append([float(o["loss_chfr"]), float(o["grad_norm"])])
        else:
            g = E.GraphedPretrainStep.__new__(E.GraphedPretrainStep)
            g.static_in, g.static_noise, g.grad_sync = pool[0].clone(), noise[0].clone(), sync
            sync.overlap = False
            for i in range(3):
                o = E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, args, grad_sync=sync, mask_noise=noise[i], augment=False)
                hist.append([float(o["loss_chfr"]), float(o["grad_norm"])])
            torch.cuda.synchronize()
            g.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.graph):
                g.out = E.step_forward_backward(m, ema, g.static_in, 200, args, grad_sync=sync, mask_noise=g.static_noise,
                                                augment=False, optimizer=opt)
            g.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.graph2, pool=g.graph.pool()):
                g.out["grad_norm"] = E.step_update(m, ema, opt)
            for i in range(3, 8):
                o = g(pool[i % 3], noise[i])
                torch.cuda.synchronize()
                hist.append([float(o["loss_chfr"]), float(o["grad_norm"])])
        res[mode] = hist
    assert all(x == x for a in res["graph"] for x in a)
    assert res["graph"] == res["eager"], res           # exact: same kernels, fixed-order reductions


@pytest.mark.parametrize("use_graphs", [False, True])
def test_segmented_ddp_step_equals_single_backward(use_graphs):
    """SegmentedDDPStep (three autograd segments, gradients stored per segment into the segment-ordered flat buffer, the
    collectives between them skipped without a process group) against the ordinary step on the same inputs: same losses, same
    gradient norm, same parameters after every step."""
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    B, steps = 8, 4
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    pool = [clouds.gaussian(B, 1024, 70 + i).cuda() for i in range(3)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(100 + i)).cuda() for i in range(steps + 3)]

    def build(segmented):
        torch.manual_seed(0)
        m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
        for mod in m.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment if segmented else None)
        return m, ema, opt

    m, ema, opt = build(False)
    ref = []
    for i in range(steps + 3):
        o = E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, args, mask_noise=noise[i], augment=False)
        ref.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
    ref_params = {k: v.detach().clone() for k, v in m.named_parameters()}

    m2, ema2, opt2 = build(True)
    rng = opt2.segment_ranges
    assert sorted(rng) == [0, 1, 2] and rng[0][0] == 0 and rng[0][1] == rng[1][0] and rng[1][1] == rng[2][0] and rng[2][1] == opt2.n
    got = []
    if use_graphs:
        # the constructor's three warm-up iterations must see the same inputs as the reference's first three steps
        seg = E.SegmentedDDPStep.__new__(E.SegmentedDDPStep)
        E.SegmentedDDPStep.__init__(seg, m2, ema2, opt2, args, pool[0], 200, warmup_iters=0, augment=False, inject_mask_noise=True,
                                    use_graphs=False)
        for i in range(3):
            o = seg(pool[i % 3].clone(), noise[i])
            got.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
        seg = E.SegmentedDDPStep(m2, ema2, opt2, args, pool[0], 200, warmup_iters=0, augment=False, inject_mask_noise=True,
                                 use_graphs=True)
        start = 3
    else:
        seg = E.SegmentedDDPStep(m2, ema2, opt2, args, pool[0], 200, warmup_iters=0, augment=False, inject_mask_noise=True,
                                 use_graphs=False)
        start = 0
    for i in range(start, steps + 3):
        o = seg(pool[i % 3].clone(), noise[i])
        torch.cuda.synchronize()
        got.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
    # Every gradient is produced by the same kernel on the same operands in both layouts (a cut tensor's gradient is one tensor
    # either way, never a sum over segments): the losses of the first step are EQUAL.  The segment-ordered flat layout changes the
    # order in which the clip's norm sums the buffer, so the norm may differ in its last bits, with it the clip factor (the norm
    # is ~10 here, above max_norm = 5), then parameters by ~1e-7 relative, a few bf16 weight shadows by one ulp, and the later
    # steps' losses at the 1e-4 level: stated bounds 1e-6 (norm, step 0), 1e-3 (later steps), 1e-4 (parameters).
    assert got[0][0] == ref[0][0] and got[0][1] == ref[0][1], (got[0], ref[0])
    assert abs(got[0][2] - ref[0][2]) <= 1e-6 * abs(ref[0][2]), (got[0], ref[0])
    for a, b in zip(got, ref):
        for x, y in zip(a, b):
            assert x == x and abs(x - y) <= 1e-3 * abs(y), (got, ref)
    worst = max(float((p.detach().float() - ref_params[k].float()).abs().max() / ref_params[k].float().abs().max().clamp_min(1e-3))
                for k, p in m2.named_parameters())
    assert worst <= 1e-4, worst


def test_segmented_ddp_step_through_rccl_group_of_one():
    """The same step with a real process group (backend nccl = RCCL, world size 1): the three async all-reduces, their waits and
    the per-step buffer broadcast are issued between the graphs; a one-rank mean all-reduce is the identity, so the trajectory
    must equal the group-less run."""
    import os
    import torch.distributed as dist
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    B = 8
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    pool = [clouds.gaussian(B, 1024, 170 + i).cuda() for i in range(2)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(200 + i)).cuda() for i in range(4)]

    def run():
        torch.manual_seed(0)
        m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
        for mod in m.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment)
        seg = E.SegmentedDDPStep(m, ema, opt, args, pool[0], 200, warmup_iters=0, augment=False, inject_mask_noise=True)
        out = []
        for i in range(4):
            o = seg(pool[i % 2].clone(), noise[i])
            torch.cuda.synchronize()
            out.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
        return out

    ref = run()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got = run()
    finally:
        dist.destroy_process_group()
    assert all(x == x for a in got for x in a)
    assert got == ref, (got, ref)                      # a one-rank mean all-reduce is the identity: exact


def test_parallel_decoders_equal_serial():
    """The loss-prediction decoder on a second stream (default) against both decoders on one stream: same losses, same gradients
    (the two branches share no reduction, so the results are bit-identical), eager and through a captured step."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
    x = torch.randn(8, 1024, 3, device="cuda") * 0.3
    noise = torch.rand(8, 64, device="cuda")
    res = {}
    was = M.PARALLEL_DECODERS
    try:
        for par in (True, False):
            M.PARALLEL_DECODERS = par
            torch.manual_seed(0)
            model = M.mae_vit_base_patch16_dec512d8b().cuda().train()
            for m in model.modules():
                if hasattr(m, "drop_prob"):
                    m.drop_prob = 0.0
            ema = E.ModelEma(model, 0.999)
            opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
            losses = []
            for _ in range(2):
                out = E.step_forward_backward(model, ema, x.clone(), 200, args, optimizer=opt, augment=False, mask_noise=noise)
                opt.gather_grads()
                losses.append((float(out["loss"]), float(out["loss_learn"])))
                g = opt.G.clone()
                E.step_update(model, ema, opt)
            torch.cuda.synchronize()
            res[par] = (losses, g, opt.P.clone())
    finally:
        M.PARALLEL_DECODERS = was
    assert res[True][0] == res[False][0]
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])


def test_inference_stack_split_equals_whole():
    """An inference-only block stack (the EMA teacher's) run as two half-batch chains on two streams (default) against the whole
    batch in one chain: every kernel works row by row, so the outputs are bit-identical."""
    from gm3d_amd import fused, models_mae_learn_loss as M
    torch.manual_seed(0)
    model = M.mae_vit_base_patch16_dec512d8b().cuda().eval()
    x = torch.randn(64, 1024, 3, device="cuda") * 0.3
    mask = torch.zeros(64, 64, dtype=torch.bool, device="cuda")
    was = fused.NOGRAD_SPLIT
    outs = {}
    try:
        for ns in (2, 1):
            fused.NOGRAD_SPLIT = ns
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                o = model(x, mask=mask, num_visible=64, need_pix_pred=False)
            torch.cuda.synchronize()
            outs[ns] = (o["loss_pred"].float().clone(), o["features"].float().clone())
    finally:
        fused.NOGRAD_SPLIT = was
    assert torch.equal(outs[2][0], outs[1][0]) and torch.equal(outs[2][1], outs[1][1])


def test_async_weight_gradients_equal_inline():
    """Inside the engine's backward the encoder's weight-gradient GEMMs run on a side stream and the decoders' are deferred to the
    same point (fused.async_wgrad); the flat gradient buffer must be what the in-line order produces, eager and graph-captured."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E, fused, models_mae_learn_loss as M
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
    x = torch.randn(8, 1024, 3, device="cuda") * 0.3
    noise = torch.rand(8, 64, device="cuda")
    res = {}
    was = fused.ASYNC_WGRAD
    try:
        for on in (True, False):
            fused.ASYNC_WGRAD = on
            torch.manual_seed(0)
            model = M.mae_vit_base_patch16_dec512d8b().cuda().train()
            for m in model.modules():
                if hasattr(m, "drop_prob"):
                    m.drop_prob = 0.0
            ema = E.ModelEma(model, 0.999)
            opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
            for _ in range(2):
                E.step_forward_backward(model, ema, x.clone(), 200, args, optimizer=opt, augment=False, mask_noise=noise)
                opt.gather_grads()
                g = opt.G.clone()
                E.step_update(model, ema, opt)
            torch.cuda.synchronize()
            assert not fused._ASYNC_WGRAD["deferred"] and fused._ASYNC_WGRAD["stream"] is None
            res[on] = (g, opt.P.clone())
    finally:
        fused.ASYNC_WGRAD = was
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_flat_gradsync_all_reduces_on_every_step(monkeypatch):
    """pretrain_step and the two-graph GraphedPretrainStep with GradSync.from_flat must issue one collective per
    bucket on EVERY step (world forced to 2, dist.all_reduce replaced by a counter)."""
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    B = 4
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    x = clouds.uniform(B, 1024, 5).cuda()
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema)
    sync = E.GradSync.from_flat(opt, bucket_bytes=32 << 20)
    nb = len(sync.buckets)
    calls = []

    class Work:
        def wait(self):
            pass

    monkeypatch.setattr(E.dist, "all_reduce", lambda t, **k: (calls.append(t.numel()), Work())[1])
    sync.world = 2
    for step in range(3):
        E.pretrain_step(m, ema, opt, x.clone(), 200, args, grad_sync=sync)
        assert len(calls) == nb * (step + 1), (step, len(calls))
    g = E.GraphedPretrainStep(m, ema, opt, args, x, 200, warmup_iters=0, grad_sync=sync)
    assert g.graph2 is not None
    del calls[:]
    for step in range(3):
        g(x)
        assert len(calls) == nb * (step + 1), (step, len(calls))
    torch.cuda.synchronize()
    assert sum(calls[:nb]) == opt.G.numel()


@pytest.mark.parametrize("warm", [0, 3])
@pytest.mark.parametrize("kind", ["single", "segmented"])
def test_replays_write_nothing_outside_the_graphs_memory(kind, warm):
    """Captured with NO eager warm-up (the first execution of the model is the capture itself) or after the production's three eager
    iterations, then replayed: memory the allocator hands
    out after the capture must stay as the test fills it.  Round 4: ModelEma settled its buffer pairs on the first update() -- inside
    the capture of the optimizer graph when nothing had run eagerly -- re-pointing the BatchNorm counters the forward graph had already
    captured; the freed counters' addresses went to the next small allocation and every replay added 1 there (a cloned loss grew by one
    ulp per replay).  The pairs are settled at optimizer construction now, and update() refuses to settle them inside a capture."""
    from types import SimpleNamespace
    from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
    from tests import clouds
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    # B = 32: every product of the step runs on a kernel of our own.  (At B <= 16 the weight gradients fall back to torch.bmm, and the
    # BLAS library cannot meet a new problem shape inside a capture: hipBLASLt ends the process with exit code 1.  Steps built with
    # their default warmup_iters meet every shape eagerly first.)
    data = clouds.gaussian(32, 1024, 77).cuda()
    torch.manual_seed(5)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    for mod in m.modules():          # (DropPath's keep-probability table is a host-to-device copy on first use: not inside a capture)
        if isinstance(mod, M.DropPath):
            mod.drop_prob = 0.0
    ema = E.ModelEma(m, 1.0)                            # decay 1 and (below) learning rate 0: parameters and teacher stay put, so every
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, weight_decay=0.0,     # replay must leave the same gradient buffer
                            segment_of=E.ddp_segment if kind == "segmented" else None)
    assert ema._pairs is not None                      # settled by the optimizer's constructor
    args.lr = 0.0
    E.adjust_learning_rate(opt, 200.0, args)
    if kind == "segmented":
        step = E.SegmentedDDPStep(m, ema, opt, args, data, 200, warmup_iters=warm, augment=False, broadcast=False, inject_mask_noise=True)
    else:
        step = E.GraphedPretrainStep(m, ema, opt, args, data, 200, warmup_iters=warm, augment=False, inject_mask_noise=True)
    noise = torch.rand(32, 64, generator=torch.Generator().manual_seed(3)).cuda()
    step(data, noise)
    torch.cuda.synchronize()
    g1, p1 = opt.G.clone(), opt.P.clone()
    # (a) stray WRITES: small and large blocks of the size classes the step itself uses, filled with a sentinel
    guards = [torch.full((n,), 12345.0, device="cuda") for n in (1, 2, 3, 8, 64, 128, 1024) for _ in range(64)]
    guards += [torch.full((1 << 20,), 12345.0, device="cuda") for _ in range(8)]
    ints = [torch.full((), 777, dtype=torch.int64, device="cuda") for _ in range(256)]
    # (b) stray READS: whatever else the allocator still holds free goes to NaN-filled blocks -- a replay that reads memory freed since
    # the capture now reads NaN instead of stale but plausible values
    poison = [torch.full((n,), float("nan"), device="cuda") for n in (1, 4, 16, 96, 384, 1536, 1 << 14, 1 << 18) for _ in range(128)]
    poison += [torch.full((1 << 24,), float("nan"), device="cuda") for _ in range(8)]
    for _ in range(3):
        out = step(data, noise)
    torch.cuda.synchronize()
    assert all(bool((g == 12345.0).all()) for g in guards)
    assert all(int(t) == 777 for t in ints)
    assert torch.equal(opt.P, p1) and torch.equal(opt.G, g1)
    assert float(out["loss"]) == float(out["loss"])
    del poison


def test_model_ema_refuses_to_settle_its_pairs_inside_a_capture():
    from gm3d_amd import engine_pretrain as E
    net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.BatchNorm1d(8)).cuda()
    ema = E.ModelEma(net, 0.99)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with pytest.raises(RuntimeError, match="prepare"):
        with torch.cuda.graph(g, stream=side):
            ema.update(net)
    ema.prepare(net)
    ema.update(net)            # eager: fine
