"""qkv projection + attention in one launch (csrc/attention.hip attn_qkv_fwd_bf16_kernel) against the two launches it replaces:
gm3d_gemm_tn_bf16_ring (same MFMA accumulation order, so the bf16 products must be BIT-identical) followed by
gm3d_attention_fwd, and against an fp64 restatement of timm Attention.forward (Point-MAE_SA3D/models/Point_MAE.py:113-122)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T", [(1, 64), (3, 25), (5, 1), (2, 33), (64, 64), (7, 63), (128, 64)])
@pytest.mark.parametrize("train", [False, True])
def test_fused_equals_two_launches(B, T, train):
    from gm3d_amd import fused, gemm
    H, C = 6, 384
    g = torch.Generator().manual_seed(B * 100 + T)
    h = torch.randn(B * T, C, generator=g).cuda().bfloat16()
    w = (torch.randn(3 * C, C, generator=g) * 0.06).cuda().bfloat16()
    assert fused.attention_qkv_supported(h, w, T, H)
    scale = 0.125
    qkv_ref = gemm.linear_tn_ring(h, w, bm=64)
    a_ref, lse_ref = fused._attention_fwd(qkv_ref, B, T, H, scale)
    a, lse, qkv = fused._attention_qkv_fwd(h, w, B, T, H, scale, want_qkv=train, want_lse=train)
    torch.cuda.synchronize()
    assert torch.equal(a, a_ref)
    if train:
        assert torch.equal(qkv, qkv_ref) and torch.equal(lse, lse_ref)
    else:
        assert lse is None and qkv is None
    # fp64 restatement from the same bf16 inputs: the band is bf16 rounding of qkv and of the probabilities
    q, k, v = (h.double() @ w.double().t()).view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B * T, C)
    err = (a.double() - ref).abs().max().item()
    assert err <= 2.0 ** -6 * max(1.0, ref.abs().max().item()), err


def test_unsupported_shapes_are_refused():
    from gm3d_amd import fused
    from gm3d_amd._capi import Gm3dError
    h = torch.zeros(65, 384, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(1152, 384, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(Gm3dError):
        fused._attention_qkv_fwd(h, w, 1, 65, 6, 0.125)
    with pytest.raises(Gm3dError):
        fused._attention_qkv_fwd(h.float(), w.float(), 1, 65, 6, 0.125)
