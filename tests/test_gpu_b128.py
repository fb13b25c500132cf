"""-m gpu: BASELINE config #2 AT ITS REAL SIZE -- B=128 clouds, N=1024, G=64, k=32, bf16, hipGraph replay: the configuration the
headline clouds/s figure of bench.py comes from (the other step tests run B=2..16).

  * FPS centres and KNN neighbourhoods of all 128 clouds bit-exact against the CPU oracle;
  * every generated mask row has exactly 39 masked / 25 visible tokens;
  * 6 replayed steps == 6 eager steps on the same inputs (the size-dependent failure mode found in round 1 -- torch's multi-block
    reduce_kernel returning stale values under replay above ~64k elements, engine_pretrain.GraphedPretrainStep -- only shows at
    full size);
  * the bf16 step's losses sit within the bf16 band of an fp32 eager step on identical inputs / noise.
Randomness is removed as in tests/test_gpu_graph.py (DropPath off, augmentation off, mask noise injected)."""
from types import SimpleNamespace

import pytest
import torch

from tests import clouds

pytestmark = pytest.mark.gpu

B = 128


def _args(bf16):
    return SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=bf16, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)


def _build(flat=True):
    from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    for mod in m.modules():
        if isinstance(mod, M.DropPath):
            mod.drop_prob = 0.0
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema) if flat else E.build_optimizer(m, lr=2e-4)
    return m, ema, opt


def test_grouping_of_128_clouds_bit_exact(oracle_ops):
    from gm3d_amd import models_mae_learn_loss as M
    x = clouds.uniform(B, 1024, 7)
    nb, cen, nbo = M.Group(64, 32)(x.cuda())
    fidx = oracle_ops.furthest_point_sample(x, 64)
    rcen = torch.gather(x, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    _, ri = oracle_ops.knn(x, rcen, 32)
    rnb, rnbo = oracle_ops.group(x, rcen, ri)
    assert torch.equal(cen.cpu(), rcen)
    assert torch.equal(nb.cpu(), rnb) and torch.equal(nbo.cpu(), rnbo)


def test_b128_bf16_graph_replay_equals_eager_and_fp32_band():
    from gm3d_amd import engine_pretrain as E
    steps = 6
    pool = [clouds.uniform(B, 1024, 150 + i).cuda() for i in range(3)]
    noise = [torch.rand(B, 64, generator=torch.Generator().manual_seed(i)).cuda() for i in range(steps + 3)]
    keys = ("loss_chfr", "loss_learn", "grad_norm")

    m, ema, opt = _build()
    eager, masks = [], []
    for i in range(steps + 3):
        o = E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, _args(True), mask_noise=noise[i], augment=False)
        eager.append([float(o[k]) for k in keys])
        masks.append(o["mask"].clone())
    for mk in masks:
        assert mk.dtype == torch.bool and mk.shape == (B, 64) and bool((mk.sum(dim=1) == 39).all())
    p_eager = opt.P.clone()
    del m, ema, opt

    m, ema, opt = _build()
    for i in range(3):      # the same three eager iterations, then capture without further warm-up
        E.pretrain_step(m, ema, opt, pool[i % 3].clone(), 200, _args(True), mask_noise=noise[i], augment=False)
    g = E.GraphedPretrainStep(m, ema, opt, _args(True), pool[0], 200, warmup_iters=0, augment=False, inject_mask_noise=True)
    got = []
    for i in range(3, steps + 3):
        o = g(pool[i % 3], noise[i])
        torch.cuda.synchronize()
        got.append([float(o[k]) for k in keys])
        assert torch.equal(o["mask"], masks[i])                 # same teacher, same noise: identical masks
    # replay == eager, EXACTLY: every kernel of the step is deterministic (own kernels with fixed-order reductions; the index
    # gathers' backward scatters over permutations, i.e. without colliding atomics) and a captured launch runs the same code on
    # the same operands (tools/eager_vs_graph_diag.py: flat gradient buffer bit-identical eager vs captured, B = 4 and 32)
    assert got == eager[3:], (got, eager[3:])
    assert torch.equal(opt.P, p_eager), float((opt.P - p_eager).abs().max())
    del g, m, ema, opt

    # fp32 eager step on the first batch: the bf16 losses of step 0 within the bf16 band (8 mantissa bits through 16 blocks)
    m, ema, opt = _build(flat=False)
    o = E.pretrain_step(m, ema, opt, pool[0].clone(), 200, _args(False), mask_noise=noise[0], augment=False)
    f32 = [float(o[k]) for k in keys]
    assert abs(eager[0][0] - f32[0]) <= 3e-2 * abs(f32[0]), (eager[0], f32)       # Chamfer loss
    assert abs(eager[0][1] - f32[1]) <= 5e-2 * abs(f32[1]), (eager[0], f32)       # ranking loss (sign decisions on near-ties)
