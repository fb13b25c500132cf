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
