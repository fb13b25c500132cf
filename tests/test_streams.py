"""gm3d_amd/streams.py: the audit that turns an unjoined side stream inside a hipGraph capture into a Python error
(the record is gpurun_out/segv.txt: SIGSEGV inside capture_end).  Ledger logic on the CPU with stand-in streams;
one GPU test that the real capture of the pretrain step raises -- and that the process survives."""
from types import SimpleNamespace

import pytest
import torch

from gm3d_amd import streams as S


class FakeStream:
    n = 0

    def __init__(self):
        FakeStream.n += 1
        self.cuda_stream = FakeStream.n
        self.waited = []

    def wait_stream(self, other):
        self.waited.append(other.cuda_stream)


@pytest.fixture
def ledger():
    origin = FakeStream()
    led = S._Ledger(origin)
    S._ledgers.append(led)
    yield origin, led
    S._ledgers.pop()


def test_fork_join_closes(ledger):
    origin, led = ledger
    a = FakeStream()
    S.fork(a, origin, who="A")
    assert [w for _, w in S.open_streams()] == ["A"]
    S.join(a, origin)
    assert S.open_streams() == [] and origin.waited == [a.cuda_stream] and a.waited == [origin.cuda_stream]


def test_nested_fork_joined_into_its_parent_reopens_the_parent(ledger):
    origin, led = ledger
    a, b = FakeStream(), FakeStream()
    S.fork(a, origin, who="A")
    S.fork(b, a, who="B")           # forked while running on A
    S.join(a, origin)               # A comes back first ...
    S.join(b, a)                    # ... then B is joined into A, which the origin no longer waits for
    left = S.open_streams()
    assert len(left) == 1 and left[0][0] is a and "B" in left[0][1]
    S.join(a, origin)
    assert S.open_streams() == []


def test_nested_fork_joined_straight_into_the_origin(ledger):
    origin, led = ledger
    a, b = FakeStream(), FakeStream()
    S.fork(a, origin, who="A")
    S.fork(b, a, who="B")
    S.join(b, origin)
    S.join(a, origin)
    assert S.open_streams() == []


def test_forking_the_origin_itself_is_not_booked(ledger):
    origin, led = ledger
    other = FakeStream()
    S.fork(origin, other, who="origin waits for a producer")
    assert S.open_streams() == []


def test_outside_a_capture_nothing_is_booked():
    a, b = FakeStream(), FakeStream()
    S.fork(a, b, who="x")
    assert S.open_streams() == [] and a.waited == [b.cuda_stream]


@pytest.mark.gpu
def test_unjoined_side_stream_in_the_step_capture_raises_and_the_process_lives():
    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from tests import clouds
    B = 4
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema)
    x = clouds.uniform(B, 1024, 5).cuda()
    rogue = torch.cuda.Stream()
    scratch = torch.zeros(1024, device="cuda")

    def leaky(model, model_ema, samples, epoch, a, **kw):
        out = E.step_forward_backward(model, model_ema, samples, epoch, a, **kw)
        if torch.cuda.is_current_stream_capturing():
            S.fork(rogue, who="test: rogue fork that nobody joins")
            with torch.cuda.stream(rogue):
                scratch.add_(1.0)
        return out

    with pytest.raises(RuntimeError, match="rogue fork that nobody joins"):
        E.GraphedPretrainStep(m, ema, opt, args, x, 200, fwd_bwd=leaky)
    torch.cuda.synchronize()
    # the audit joined the stream before the capture ended: the runtime is intact and the same model captures cleanly
    g = E.GraphedPretrainStep(m, ema, opt, args, x, 200)
    o = g(x)
    torch.cuda.synchronize()
    assert float(o["loss"]) == float(o["loss"])
    assert S.open_streams() == []


@pytest.mark.gpu
def test_a_cached_weight_cast_is_waited_for_across_streams():
    """fused.weight_cache.get casts a weight once and hands the copy to whoever asks next.  The parallel inference chains of run_stack ask
    from DIFFERENT streams a few microseconds apart: the second one must wait for the casting launch (round 4: the teacher's forward as
    two chains differed from the single chain once in a full test run beside a loader process -- eager mode with un-pinned weights)."""
    import torch
    from gm3d_amd import fused
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    w = torch.randn(2048, 2048, device="cuda")
    big = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        for _ in range(20):                       # keeps stream a busy: the cast below is queued behind ~100 ms of products
            big = big @ big * 1e-4
        c1 = fused.weight_cache.get(w, torch.bfloat16)
    with torch.cuda.stream(b):                    # no dependency on stream a except through the cache
        c2 = fused.weight_cache.get(w, torch.bfloat16)
        got = c2.float().abs().sum()
    torch.cuda.synchronize()
    assert c2 is c1
    assert float(got) == float(w.bfloat16().float().abs().sum())
