"""Small engine-side kernels (csrc/embed.hip) against the tensor expressions they replace."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ema_counters_kernel_equals_the_tensor_expression():
    """gm3d_ema_counters == e.copy_(e * decay + (1.0 - decay) * m) on int64 counters (fp32 arithmetic with the Python scalars rounded to
    fp32, truncating conversion), for the decays of the schedule and counters up to a long run's size."""
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    g = torch.Generator(device="cuda").manual_seed(1)
    for decay in (0.999, 0.9995, 0.9999, 0.99):
        e = torch.randint(0, 3_000_000, (37,), device="cuda", generator=g)
        m = e + torch.randint(0, 5000, (37,), device="cuda", generator=g)
        want = e.clone()
        want.copy_(want * decay + (1.0 - decay) * m)
        check(lib.gm3d_ema_counters(_ptr(e), _ptr(m), 37, decay, 1.0 - decay, _stream()), "ema")
        assert torch.equal(e, want)
