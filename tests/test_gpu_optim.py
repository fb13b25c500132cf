"""-m gpu: FlatAdamWEma (one-pass clip + AdamW + EMA + bf16 shadows, gm3d_amd/optim.py) against
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (reference parameter groups) + the EMA formula, over several steps."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_flat_optimizer_matches_torch():
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd.fused import weight_cache
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.LayerNorm(53), torch.nn.Linear(53, 11), torch.nn.BatchNorm1d(11)).cuda()
    net.register_parameter("mask_token", torch.nn.Parameter(torch.randn(1, 1, 7, device="cuda")))
    ref = copy.deepcopy(net)
    ema = E.ModelEma(net, decay=0.9)
    ema_ref = copy.deepcopy(ref).eval()
    opt = E.build_optimizer(net, lr=3e-3, weight_decay=0.05, flat=True, model_ema=ema, clip_grad=0.5)
    opt_ref = torch.optim.AdamW(E.add_weight_decay(ref, 0.05), lr=3e-3)
    assert opt.n % 4 == 0 and opt.n_decay % 4 == 0
    x = torch.randn(64, 37, device="cuda")
    for step in range(5):
        # identical gradients on both sides (Adam's m/(sqrt(v)+eps) turns 1e-7 gradient noise on near-zero entries into
        # visible parameter differences, which would test the model's conditioning, not the optimizer arithmetic)
        opt_ref.zero_grad(set_to_none=True)
        (ref(x * (1 + step)).pow(2).mean() * 50 + ref.mask_token.sum()).backward()
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            net(x * (1 + step))          # same BatchNorm running-stat update as the reference side
        for a, b in zip(net.parameters(), ref.parameters()):
            a.grad = b.grad.clone()
        gn_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt_ref.step()
        with torch.no_grad():
            for (k, e), (_, p) in zip(ema_ref.state_dict().items(), ref.state_dict().items()):
                e.copy_((e * 0.9 + 0.1 * p).to(e.dtype))
        gn = E.step_update(net, ema, opt)
        assert abs(float(gn) - float(gn_ref)) <= 1e-5 * float(gn_ref)
        for (k, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
            assert (a - b).abs().max() <= 2e-6 * b.abs().max() + 1e-7, (step, k)
        for (k, a), (_, b) in zip(ema.ema.state_dict().items(), ema_ref.state_dict().items()):
            assert (a.float() - b.float()).abs().max() <= 2e-6 * b.float().abs().max() + 1e-7, (step, k)
    # bf16 shadows follow the masters; state dict has the AdamW shape
    w = net[0].weight
    assert torch.equal(weight_cache.get(w, torch.bfloat16), w.detach().bfloat16())
    tw = ema.ema[0].weight
    assert torch.equal(weight_cache.get(tw, torch.bfloat16), tw.detach().bfloat16())
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 5.0
    E.adjust_learning_rate(opt, 10.0, type("A", (), dict(lr=1e-3, min_lr=0.0, warmup_epochs=40, epochs=400)))
    assert abs(float(opt.lr_dev) - 2.5e-4) < 1e-9


def test_direct_weight_gradients_land_in_flat_buffer():
    """The block stacks' batched weight-gradient GEMMs write into the flat gradient buffer (same-kind weights adjacent in the
    layout): after backward the parameters' .grad ARE their slots, gather_grads copies only the rest, and the gradients equal
    those of the same step with the direct path off."""
    import gm3d_amd.optim as O
    from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
    from types import SimpleNamespace
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
    x = torch.randn(8, 1024, 3, device="cuda") * 0.3
    noise = torch.rand(8, 64, device="cuda")
    res = {}
    for direct in (True, False):
        O.ENABLE_DIRECT_WGRAD = direct
        torch.manual_seed(0)
        model = M.mae_vit_base_patch16_dec512d8b().cuda().train()
        for m in model.modules():
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0
        ema = E.ModelEma(model, 0.999)
        opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
        E.step_forward_backward(model, ema, x.clone(), 200, args, optimizer=opt, augment=False, mask_noise=noise)
        named = dict(model.named_parameters())
        w = named["blocks.blocks.5.mlp.fc1.weight"]
        slot = O.grad_slots.get(w)
        assert (w.grad.data_ptr() == slot.data_ptr()) == direct
        # same-kind weights of a stack are adjacent in the layout
        w6 = O.grad_slots.get(named["blocks.blocks.6.mlp.fc1.weight"])
        assert w6.data_ptr() == slot.data_ptr() + slot.numel() * 4
        opt.gather_grads()
        res[direct] = opt.G.clone()
        opt._gathered = False
    O.ENABLE_DIRECT_WGRAD = True
    a, b = res[True].double(), res[False].double()
    assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())


def test_state_dict_round_trip_by_parameter_name_and_stale_gradient_guard():
    """(1) the optimizer entry of a checkpoint carries parameter names; a FlatAdamWEma with ANOTHER layout (segment
    order) restores every moment to the right parameter, a mismatching file is refused; (2) step() after zero_grad() with no
    backward, or with a parameter that received no gradient, uses zeros -- never the previous step's gradients."""
    from gm3d_amd import engine_pretrain as E
    torch.manual_seed(0)

    def net():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.LayerNorm(16), torch.nn.Linear(16, 16)).cuda()

    a = net()
    oa = E.build_optimizer(a, lr=1e-3, flat=True)
    x = torch.randn(8, 16, device="cuda")
    for _ in range(2):
        oa.zero_grad()
        a(x).pow(2).mean().backward()
        oa.step()
    sd = oa.state_dict()
    assert sd["param_names"] == [n for n, _ in oa._named]
    b = net()
    ob = E.build_optimizer(b, lr=1e-3, flat=True, segment_of=lambda n: 0 if n.startswith("2.") else 1)   # "2.*" first: another order
    assert [n for n, _ in ob._named] != [n for n, _ in oa._named]
    ob.load_state_dict(sd)
    ma = {n: oa.state[p]["exp_avg"] for n, p in oa._named}
    for n, p in ob._named:
        assert torch.equal(ob.state[p]["exp_avg"], ma[n]), n
    assert float(ob.step_dev) == 2.0
    bad = dict(sd, param_names=["x." + n for n in sd["param_names"]])
    with pytest.raises(ValueError):
        ob.load_state_dict(bad)
    # (2) a step without gradients moves parameters only by weight decay (Adam's moments decay towards zero)
    oa.zero_grad()
    a(x).pow(2).mean().backward()
    oa.step()
    assert float(oa.G.abs().max()) > 0
    oa.zero_grad()
    oa.step()                       # no backward in between
    assert float(oa.G.abs().max()) == 0.0
    oa.zero_grad()
    a[0](x).pow(2).mean().backward()        # only the first layer receives a gradient
    oa.step()
    slot = {id(p): v for p, v in oa.flat_grad_views()}
    assert float(slot[id(a[2].weight)].abs().max()) == 0.0 and float(slot[id(a[0].weight)].abs().max()) > 0


def test_ema_weight_survives_graph_replays_at_another_decay():
    """eager step at decay A, captured steps at decay B replayed, eager step at decay A again -- the EMA
    parameters must equal those of an all-eager twin (the device-side EMA weight must not be left at the graph's value)."""
    from gm3d_amd import engine_pretrain as E
    torch.manual_seed(0)

    def make():
        torch.manual_seed(1)
        net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.LayerNorm(64), torch.nn.Linear(64, 16)).cuda()
        ema = E.ModelEma(net, decay=0.5)
        opt = E.build_optimizer(net, lr=1e-2, weight_decay=0.05, flat=True, model_ema=ema, clip_grad=1.0)
        return net, ema, opt

    grads = [torch.randn(64 * 32 + 64 + 64 + 64 + 16 * 64 + 16, device="cuda") for _ in range(6)]
    decays = [0.5, 0.9, 0.9, 0.9, 0.5, 0.9]

    def run(graphed):
        net, ema, opt = make()
        assert opt.G.numel() >= grads[0].numel()
        static_g = torch.zeros_like(opt.G)
        graph = None
        for i, (g, d) in enumerate(zip(grads, decays)):
            ema.decay = d
            static_g.zero_()
            static_g[:g.numel()].copy_(g)
            if graphed and d == 0.9:
                if graph is None:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        opt.G.copy_(static_g)
                        opt.mark_grads_filled()
                        E.step_update(net, ema, opt)
                graph.replay()                  # (a capture executes nothing: the first captured step runs as a replay too)
            else:
                opt.G.copy_(static_g)
                opt.mark_grads_filled()
                E.step_update(net, ema, opt)
        torch.cuda.synchronize()
        return opt.P.clone(), opt.E.clone()

    p0, e0 = run(False)
    p1, e1 = run(True)
    assert torch.equal(p0, p1)
    assert torch.equal(e0, e1)
