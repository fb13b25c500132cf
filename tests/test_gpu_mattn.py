"""-m gpu: masked attention of the hierarchical encoder (gm3d_attention_masked_fwd/bwd through ops.attention_masked) against an
fp64 torch restatement: symmetric random masks (a radius-style mask on random centres + padding of a variable-length prefix),
head_dim 16/32/64, up to 512 tokens; fully masked (padding) queries give zero output and zero gradient.
f32 mode: 1e-5 relative; bf16 mode: bf16 rounding of P and the outputs (tolerance at the asserts)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(B, T, seed, radius=0.9, pad=True):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(B, T, 3, generator=g)
    m = torch.cdist(c, c) >= radius                                     # symmetric, diagonal False
    if pad:
        n = torch.randint(max(T // 2, 1), T + 1, (B,), generator=g)
        valid = torch.arange(T).unsqueeze(0) < n.unsqueeze(1)           # (B,T)
        m = m | ~valid.unsqueeze(1) | ~valid.unsqueeze(2)
    return m


def _ref(qkv, mask, H, hd, scale):
    B, T, _ = qkv.shape
    q, k, v = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * scale
    if mask is not None:
        s = s.masked_fill(mask.unsqueeze(1), float("-inf"))
    p = torch.nan_to_num(s.softmax(dim=-1), nan=0.0)                    # a row with no allowed key: zeros
    return (p @ v).transpose(1, 2).reshape(B, T, H * hd)


CASES = [(13, 64), (64, 64), (104, 32), (128, 32), (300, 16), (512, 16), (33, 16), (1, 32), (200, 32)]


@pytest.mark.parametrize("T,hd", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_masked_attention(T, hd, dtype):
    from gm3d_amd import ops
    torch.manual_seed(T * 7 + hd)
    B, H = 3, 6
    scale = hd ** -0.5
    mask = _mask(B, T, seed=T + hd).cuda()
    assert torch.equal(mask, mask.transpose(1, 2))
    bits = ops.pack_mask(mask)
    # the packing itself: bit j&31 of word j>>5
    j = min(T - 1, 37)
    assert torch.equal(((bits[:, :, j >> 5].long() >> (j & 31)) & 1).bool(), mask[:, :, j])
    qkv = (torch.randn(B, T, 3 * H * hd, device="cuda") * 1.2).to(dtype).requires_grad_(True)
    w = torch.randn(B, T, H * hd, device="cuda").to(dtype)
    q64 = qkv.detach().double().requires_grad_(True)
    ref = _ref(q64, mask, H, hd, scale)
    (ref * w.double()).sum().backward()
    out = ops.attention_masked(qkv, bits, H, scale)
    (out.float() * w.float()).sum().backward()
    tol_o, tol_g = (1e-5, 1e-5) if dtype == torch.float32 else (2e-2, 3e-2)
    assert (out.double() - ref).abs().max() <= tol_o * ref.abs().max()
    assert (qkv.grad.double() - q64.grad).abs().max() <= tol_g * q64.grad.abs().max()
    dead = mask.all(dim=2)                                               # padding queries
    if dead.any():
        assert float(out[dead].abs().max()) == 0.0
        g = qkv.grad.reshape(B, T, 3, H, hd)
        assert float(g[dead].abs().max()) == 0.0


def test_masked_attention_without_mask_equals_plain_attention():
    from gm3d_amd import ops
    torch.manual_seed(3)
    B, T, H = 4, 64, 6
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
    a = ops.attention(qkv, H, 0.125)
    b = ops.attention_masked(qkv, None, H, 0.125)
    assert (a.float() - b.float()).abs().max() <= 2e-2 * a.float().abs().max()
    with pytest.raises(RuntimeError):
        ops.attention_masked(torch.randn(1, 8, 3 * 6 * 24, device="cuda"), None, 6, 0.2)      # head_dim 24
