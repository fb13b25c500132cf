"""Where the data-parallel layout's overhead goes, with ONE rank over RCCL: the four graphs of SegmentedDDPStep timed one at a time
(replayed back to back, no collectives), their sum, the whole step with and without its three collectives, and the single-graph step.
    python tools/seg_timeline.py [steps]"""
import os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29656")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
pool = [bench.make_clouds(128, 1024, 1234 + 1000 * i, dev) for i in range(4)]


def build(segmented):
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b(norm_pix_loss=False).to(dev).train()
    ema = E.ModelEma(m, decay=E.ema_decay_for_epoch(200))
    opt = E.build_optimizer(m, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema, segment_of=E.ddp_segment if segmented else None)
    E.adjust_learning_rate(opt, 200.0, args)
    if segmented:
        return E.SegmentedDDPStep(m, ema, opt, args, pool[0], 200)
    return E.GraphedPretrainStep(m, ema, opt, args, pool[0], 200)


def timed(fn, n=steps, warm=10):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


seg = build(True)
whole = timed(lambda i: seg(pool[i % 4]))
parts = [timed(lambda i, g=g: g.replay()) for g in seg.graphs]
seg._reduce = lambda k: None
bare = timed(lambda i: seg(pool[i % 4]))
del seg
one = build(False)
single = timed(lambda i: one(pool[i % 4]))
print("single graph            %.3f ms" % single)
print("four graphs + 3 RCCL    %.3f ms" % whole)
print("four graphs, no RCCL    %.3f ms" % bare)
print("graphs alone            %s  sum %.3f ms" % (" ".join("%.3f" % p for p in parts), sum(parts)))
