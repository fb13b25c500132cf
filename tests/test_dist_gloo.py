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
