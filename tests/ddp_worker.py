"""One rank of the two-rank data-parallel test of the REAL model (tests/test_gpu_ddp.py): both ranks share cuda:0, backend gloo.

    python -m tests.ddp_worker --rank R --world 2 --port P --out DIR

Started by tests/conftest.py from a parent process that has not touched the GPU.  Per mode (SegmentedDDPStep eager, then
hipGraph-captured) the rank writes DIR/<mode>_rank<R>.pt with:
  g_local   flat gradient buffer of this rank's shard alone (segments run without collectives, nothing updated)
  g_avg     flat gradient buffer after the three collectives of step 1
  params    flat parameter buffer after 3 steps;  ema: the teacher's
  buf_pre / buf_post   BatchNorm running_mean of the student before / after an explicit broadcast_buffers()
  losses    per step
Models are initialised from a RANK-DEPENDENT seed (the reference seeds with args.seed + rank,
Point-MAE_SA3D/main_pretrain_multi_gpu.py:175-176): the constructor's broadcast has to make them equal.
"""
import argparse
import os
import sys
import traceback
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(rank, world, port, out, B=4, steps=3):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    data = [clouds.gaussian(B * world, 1024, 300 + i) for i in range(steps + 1)]
    noise = [torch.rand(B * world, 64, generator=torch.Generator().manual_seed(400 + i)) for i in range(steps + 1)]
    ids = E.shard_for_rank(B * world, rank, world, shuffle=False)
    for mode in ("eager", "graph"):
        torch.manual_seed(100 + rank)                       # rank-dependent initial weights: the broadcast must fix them
        m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
        for mod in m.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
        ema = E.ModelEma(m, 0.999)
        opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment)
        E.adjust_learning_rate(opt, 200.0, args)
        seg = E.SegmentedDDPStep(m, ema, opt, args, data[0][ids].cuda(), 200, warmup_iters=0, augment=False,
                                 inject_mask_noise=True, use_graphs=(mode == "graph"))
        p_start = opt.P.clone()
        # this rank's shard alone: the three segments without their collectives, run EAGERLY in both modes -- the captured
        # step of graph mode is compared with eager execution, not with itself
        seg.static_noise.copy_(noise[0][ids].cuda())
        bufs = [t.detach().clone() for t in m.buffers()]
        seg._phase1(data[0][ids].cuda())
        seg._phase2()
        seg._phase3()
        seg._cut1 = seg._cut2 = seg._cut3 = None
        with torch.no_grad():                               # the extra forward moved the BatchNorm running statistics: roll back
            for t, v in zip(m.buffers(), bufs):
                t.copy_(v)
        torch.cuda.synchronize()
        g_local = opt.G.clone()
        losses, g_avg = [], None
        for i in range(steps):
            o = seg(data[i][ids].cuda(), noise[i][ids].cuda())
            torch.cuda.synchronize()
            if i == 0:
                g_avg = opt.G.clone()
            losses.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
        bn = m.encoder.second_conv[1]
        buf_pre = bn.running_mean.clone()
        E.broadcast_buffers(m)
        torch.cuda.synchronize()
        torch.save({"p_start": p_start.cpu(), "g_local": g_local.cpu(), "g_avg": g_avg.cpu(), "params": opt.P.cpu(),
                    "ema": opt.E.cpu(), "buf_pre": buf_pre.cpu(), "buf_post": bn.running_mean.cpu(), "losses": losses,
                    "segments": {k: list(v) for k, v in opt.segment_ranges.items()},
                    "names": [n for n, _ in opt._named], "offs": list(opt._offs)},
                   os.path.join(out, "%s_rank%d.pt" % (mode, rank)))
        del seg, opt, ema, m
        torch.cuda.empty_cache()
        dist.barrier()
    # the drop-in epoch loop on two ranks: eager warm-up iterations through GradSync.from_flat (created by train_one_epoch), then the
    # captured SegmentedDDPStep; each rank feeds its own shard of every batch
    torch.manual_seed(100 + rank)
    m = M.mae_vit_base_patch16_dec512d8b().cuda()
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment)
    eargs = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40,
                            learning_loss=True)
    loader = [clouds.gaussian(B * world, 1024, 500 + i)[ids] for i in range(7)]
    E._warm.clear()
    stats = E.train_one_epoch(m, loader, opt, torch.device("cuda"), 200, None, args=eargs, model_ema=ema, print_freq=100)
    torch.cuda.synchronize()
    torch.save({"params": opt.P.cpu(), "ema": opt.E.cpu(), "stats": {k: float(v) for k, v in stats.items()}},
               os.path.join(out, "epoch_rank%d.pt" % rank))
    dist.barrier()
    full_size_leg(rank, world, out, E, M, clouds, args)
    dist.barrier()
    dist.destroy_process_group()


def full_size_leg(rank, world, out, E, M, clouds, args, B=128, steps=3):
    """BASELINE config #3 at its per-rank size: B = 128 clouds per rank, the captured four-graph SegmentedDDPStep, two ranks.  The flat
    buffers are 147 MB each: the comparisons are made HERE (the ranks exchange their eagerly computed shard gradients over gloo) and
    only verdicts, hashes and losses are written."""
    import hashlib
    import torch
    import torch.distributed as dist
    digest = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
    data = [clouds.gaussian(B * world, 1024, 900 + i) for i in range(steps + 1)]
    noise = [torch.rand(B * world, 64, generator=torch.Generator().manual_seed(950 + i)) for i in range(steps + 1)]
    ids = E.shard_for_rank(B * world, rank, world, shuffle=False)
    torch.manual_seed(100 + rank)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    for mod in m.modules():
        if isinstance(mod, M.DropPath):
            mod.drop_prob = 0.0
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment)
    E.adjust_learning_rate(opt, 200.0, args)
    seg = E.SegmentedDDPStep(m, ema, opt, args, data[0][ids].cuda(), 200, warmup_iters=0, augment=False, inject_mask_noise=True,
                             use_graphs=True)
    seg.static_noise.copy_(noise[0][ids].cuda())
    bufs = [t.detach().clone() for t in m.buffers()]
    seg._phase1(data[0][ids].cuda())          # this rank's shard alone, EAGERLY, no collectives
    seg._phase2()
    seg._phase3()
    seg._cut1 = seg._cut2 = seg._cut3 = None
    with torch.no_grad():
        for t, v in zip(m.buffers(), bufs):
            t.copy_(v)
    torch.cuda.synchronize()
    g_local = opt.G.detach().cpu().clone()
    losses, g_avg = [], None
    for i in range(steps):
        o = seg(data[i][ids].cuda(), noise[i][ids].cuda())
        torch.cuda.synchronize()
        if i == 0:
            g_avg = opt.G.detach().cpu().clone()
        losses.append([float(o["loss_chfr"]), float(o["loss_learn"]), float(o["grad_norm"])])
    both = [torch.empty_like(g_local) for _ in range(world)]
    dist.all_gather(both, g_local)
    want = (both[0] + both[1]) / 2
    offs = list(opt._offs) + [opt.G.numel()]          # a failure names its parameters and segments (diagnosis without a re-run)
    worst = sorted(((float((g_avg[o:e] - want[o:e]).abs().max()), float(want[o:e].abs().max()), n)
                    for (n, _), o, e in zip(opt._named, offs[:-1], offs[1:]) if not torch.equal(g_avg[o:e], want[o:e])), reverse=True)
    seg_equal = {int(k): bool(torch.equal(g_avg[lo:hi], want[lo:hi])) for k, (lo, hi) in opt.segment_ranges.items()}
    torch.save({"avg_equals_mean": bool(torch.equal(g_avg, want)), "max_dev": float((g_avg - want).abs().max()),
                "worst": worst[:8], "n_differ": len(worst), "segments_equal": seg_equal,
                "local_equals_avg_here": bool(torch.equal(g_avg, g_local)),
                "shards_differ": not torch.equal(both[0], both[1]), "g_avg_hash": digest(g_avg), "params_hash": digest(opt.P),
                "ema_hash": digest(opt.E), "losses": losses, "batch_per_rank": B},
               os.path.join(out, "full_rank%d.pt" % rank))
    del seg, opt, ema, m
    torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    try:
        run(a.rank, a.world, a.port, a.out)
    except BaseException:
        with open(os.path.join(a.out, "error_rank%d.txt" % a.rank), "w") as f:
            f.write(traceback.format_exc())
        raise
