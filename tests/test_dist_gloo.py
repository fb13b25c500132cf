"""CPU, world_size 2, gloo: the data-parallel gradient path (GradSync buckets + buffer broadcast) gives the
same parameters as one process on the concatenated batch.  (BatchNorm is kept out of the equivalence model:
the reference uses per-rank batch statistics)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                               torch.nn.Linear(64, 8))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gm3d_amd.engine_pretrain import GradSync, broadcast_buffers, shard_for_rank
    net = _net()
    sync = GradSync(net.parameters(), bucket_bytes=8 * 1024)      # several buckets
    assert len(sync.buckets) >= 3
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(1)
    data, target = torch.randn(32, 16, generator=g), torch.randn(32, 8, generator=g)
    for step in range(3):
        ids = shard_for_rank(32, rank, world, epoch=step, seed=0)
        sync.zero_grad()
        loss = torch.nn.functional.mse_loss(net(data[ids]), target[ids])
        loss.backward()
        sync.finish()
        opt.step()
    from gm3d_amd.validate import gather_tensor
    gathered = gather_tensor(torch.full((2, 3), float(rank)))          # validation all-gather (dist_utils.gather_tensor)
    assert gathered.shape == (4, 3) and gathered[:2].eq(0).all() and gathered[2:].eq(1).all()
    bn = torch.nn.BatchNorm1d(4)
    bn.running_mean.fill_(float(rank + 1))
    broadcast_buffers(bn)
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()], bn.running_mean.numpy().copy()))  # numpy: no shm handles
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_equals_single_process():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: the mean loss over the full batch == average of the two shard-mean gradients
    net = _net()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(1)
    data, target = torch.randn(32, 16, generator=g), torch.randn(32, 8, generator=g)
    for step in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(data), target).backward()
        opt.step()
    for a, b, c in zip(res[0][1], res[1][1], net.parameters()):
        assert (a == b).all()                                      # ranks stay bit-identical
        assert torch.allclose(torch.from_numpy(a), c.detach(), rtol=1e-5, atol=1e-6)
    assert (res[0][2] == res[1][2]).all() and float(res[1][2][0]) == 1.0   # rank 0's buffers win


def _seeded_net(seed):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.BatchNorm1d(32), torch.nn.GELU(), torch.nn.Linear(32, 8))
    net[1].running_mean.normal_()           # buffers differ per seed as well
    net[1].eval()                           # running statistics: the equivalence with one process holds exactly
    return net


def _worker_seeded_accum(rank, world, port, q):
    """Ranks start from DIFFERENT seeds (the reference seeds with args.seed + rank, P/main_pretrain_multi_gpu.py:175-176) and train
    with accum_iter = 2 through the engine's own backward_and_collect."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gm3d_amd import engine_pretrain as E
    net = _seeded_net(10 + rank)
    sync = E.GradSync(net.parameters(), bucket_bytes=2 * 1024, model=net)      # broadcasts rank 0's parameters AND buffers
    start = [p.detach().numpy().copy() for p in net.parameters()] + [net[1].running_mean.numpy().copy()]
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    data, target = torch.randn(64, 16, generator=g), torch.randn(64, 8, generator=g)
    accum, calls = 2, []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    for it in range(4):                                         # two accumulation windows
        first, last = it % accum == 0, (it + 1) % accum == 0
        rows = torch.arange(16 * it, 16 * it + 16)[rank::world]             # this rank's half of micro-batch `it`
        loss = torch.nn.functional.mse_loss(net(data[rows]), target[rows]) / accum
        was = sync.overlap
        E.backward_and_collect(loss, net, None, sync, accum, first, last)
        sync.overlap = was
        if last:
            sync.finish()
            opt.step()
    dist.all_reduce = real
    q.put((rank, start, [p.detach().numpy().copy() for p in net.parameters()], len(calls), len(sync.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_dependent_seeds_and_accumulation_window():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_seeded_accum, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _seeded_net(10)                                       # rank 0's seed
    for a, b, c in zip(res[0][1], res[1][1], list(ref.parameters()) + [ref[1].running_mean]):
        assert (a == b).all() and (a == c.detach().numpy()).all()           # the constructor's broadcast: rank 0's state everywhere
    opt = torch.optim.SGD(ref.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    data, target = torch.randn(64, 16, generator=g), torch.randn(64, 8, generator=g)
    for w in range(2):                                          # one step per window on its 32 samples
        opt.zero_grad()
        torch.nn.functional.mse_loss(ref(data[32 * w:32 * w + 32]), target[32 * w:32 * w + 32]).backward()
        opt.step()
    for a, b, c in zip(res[0][2], res[1][2], ref.parameters()):
        assert (a == b).all()
        assert torch.allclose(torch.from_numpy(a), c.detach(), rtol=1e-5, atol=1e-6)
    # collectives only at the two window ends: one per bucket each
    assert res[0][3] == 2 * res[0][4] and res[0][4] >= 2


def test_flat_gradsync_issues_every_bucket_on_every_finish(monkeypatch):
    """with the flat layout nothing calls GradSync.zero_grad(), so finish() itself must re-arm the buckets --
    otherwise only the first step of a data-parallel run is all-reduced and the replicas drift apart silently."""
    from gm3d_amd import engine_pretrain as E

    class Opt:          # the part of FlatAdamWEma that from_flat reads
        def __init__(self):
            self._params = [torch.nn.Parameter(torch.zeros(n)) for n in (700, 300, 1000, 48)]
            self._offs = [0, 700, 1000, 2000]
            self.G = torch.zeros(2048)

        def flat_grad_views(self):
            return [(p, self.G[o:o + p.numel()]) for p, o in zip(self._params, self._offs)]

    calls = []

    class Work:
        def wait(self):
            pass

    monkeypatch.setattr(E.dist, "all_reduce", lambda t, **k: (calls.append(t.numel()), Work())[1])
    sync = E.GradSync.from_flat(Opt(), bucket_bytes=4 * 512)
    sync.world = 2
    assert len(sync.buckets) == 4
    for step in range(4):
        sync.finish()
        assert len(calls) == 4 * (step + 1), (step, calls)
    assert sum(calls[:4]) == 2048
