"""-m gpu: the hand-written bf16 MFMA GEMM (gm3d_gemm_tn_bf16) against an fp32 matmul of the same bf16 operands.
Tolerance: one bf16 rounding of the result (2^-8 relative) plus fp32 accumulation-order noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(8192, 384, 1152), (3200, 1536, 384), (8192, 384, 1536), (3200, 384, 384), (100, 64, 128),
                                   (129, 128, 256), (4096, 512, 384), (1, 1152, 384), (8200, 384, 384), (8321, 1536, 384), (16384, 128, 256)])
@pytest.mark.parametrize("bias", [False, True])
def test_gemm_tn(M, K, N, bias):
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) if bias else None
    assert gemm.supported(x, w)
    y = gemm.linear_tn(x, w, b)
    ref = x.float() @ w.float().t()
    if bias:
        ref = ref + b
    err = (y.float() - ref).abs()
    assert float((err / (ref.abs() + 1.0)).max()) <= 6e-3
    assert float(err.mean()) <= 3e-3 * float(ref.abs().mean())
    assert torch.equal(y, ref.bfloat16()) or float((y.float() - ref.bfloat16().float()).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())


def test_gemm_strided_operands_and_limits():
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib
    big = torch.randn(500, 1152, device="cuda").bfloat16()
    x = big[:, 384:768]                                    # row pitch 1152, K = 384
    w = (torch.randn(256, 384, device="cuda") / 20).bfloat16()
    assert gemm.supported(x, w)
    y = gemm.linear_tn(x, w)
    assert float((y.float() - x.float() @ w.float().t()).abs().max()) <= 0.05
    out = torch.zeros(500, 512, device="cuda", dtype=torch.bfloat16)
    gemm.linear_tn(x, w, out=out[:, 256:])                 # strided output
    assert torch.equal(out[:, 256:], y) and float(out[:, :256].abs().max()) == 0.0
    assert not gemm.supported(x, w[:100])                  # N % 128
    assert lib.gm3d_gemm_tn_bf16(1, 1, None, 1, 8, 100, 64, 64, 64, 100, None) == -2


@pytest.mark.parametrize("M,with_f", [(8192, True), (3200, False), (77, True)])
def test_gemm_gelu_epilogue(M, with_f):
    """fc1 + bias + GELU fused: F = bf16(x @ w^T) (no bias), G = GELU(F + bias) -- equal to the two-kernel path
    (library GEMM, then gm3d_bias_gelu_fwd) up to the GEMMs' accumulation order."""
    from gm3d_amd import gemm, fused
    K, N = 384, 1536
    g = torch.Generator(device="cuda").manual_seed(M)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.3
    f_out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if with_f else None
    f, gg = gemm.linear_gelu(x, w, b, f_out=f_out)
    fref = (x.float() @ w.float().t())
    if with_f:
        assert float((f.float() - fref).abs().max()) <= 2.0 ** -7 * float(fref.abs().max())
        two = fused.bias_gelu_fwd(f, b, torch.bfloat16)            # same f -> must agree to the last bit
        assert torch.equal(two, gg)
    gref = torch.nn.functional.gelu(fref.bfloat16().float() + b)
    assert float((gg.float() - gref).abs().max()) <= 2.0 ** -6 * float(gref.abs().max()) + 1e-3


@pytest.mark.parametrize("groups,K,N,after,rows", [(8192, 128, 256, False, True), (8192, 512, 384, True, False), (37, 128, 256, False, True),
                                                   (5, 512, 384, True, False)])
@pytest.mark.parametrize("on_dma", [True, False])
def test_gemm_pool_epilogue(groups, K, N, after, rows, on_dma, monkeypatch):
    """conv + max over the 32 rows of each group in the GEMM epilogue == the same GEMM followed by gm3d_group_max_fwd; on both
    kernels that carry the epilogue (csrc/gemm_dma.hip: the default since round 3; csrc/gemm.hip)."""
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    monkeypatch.setattr(gemm, "POOL_ON_DMA", on_dma)
    M = groups * 32
    g = torch.Generator(device="cuda").manual_seed(groups + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.2
    full, pooled, arg = gemm.linear_pool(x, w, b, bias_after_pool=after, want_rows=rows)
    z = gemm.linear_tn(x, w, None if after else b)             # same kernel, same accumulation order: bit-identical rows
    if rows:
        assert torch.equal(full, z)
    want = torch.empty(groups, N, device="cuda", dtype=torch.bfloat16)
    warg = torch.empty(groups, N, device="cuda", dtype=torch.uint8)
    check(lib.gm3d_group_max_fwd(_ptr(z), _ptr(b) if after else None, _ptr(want), _ptr(warg), groups, 32, N, 1, _stream()), "gmax")
    assert torch.equal(pooled, want) and torch.equal(arg, warg)


@pytest.mark.parametrize("M", [8192, 3200, 200])
def test_gemm_gelu_bwd_epilogue(M):
    """fc2 input gradient + GELU backward + bias-gradient partials in one launch == own GEMM then gm3d_bias_gelu_bwd."""
    from gm3d_amd import gemm, fused
    C, Hd = 384, 1536
    g = torch.Generator(device="cuda").manual_seed(M + 1)
    d_o = torch.randn(M, C, device="cuda", generator=g).bfloat16()
    w2 = (torch.randn(C, Hd, device="cuda", generator=g) / Hd ** 0.5).bfloat16()          # fc2.weight (out=C, in=Hd)
    f = torch.randn(M, Hd, device="cuda", generator=g).bfloat16()
    b1 = torch.randn(Hd, device="cuda", generator=g) * 0.3
    w2t = gemm.stacked_transpose([w2])[0]
    assert w2t.shape == (Hd, C) and torch.equal(w2t, w2.t())
    df = torch.empty(M, Hd, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(gemm.tile_rows(M), Hd, device="cuda")
    gemm.linear_gelu_bwd(d_o, w2t, f, b1, df, part)
    dg = gemm.linear_tn(d_o, w2t)                                   # same kernel: identical accumulation
    want, db = fused.bias_gelu_bwd(dg, f, b1, torch.bfloat16)
    assert torch.equal(df, want)
    got_db = part.sum(0)
    assert float((got_db - db).abs().max()) <= 1e-4 * float(db.abs().max()) + 1e-4
    ref = (d_o.float() @ w2.float())
    assert float((dg.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())


def test_stacked_transpose_strided_view():
    from gm3d_amd import gemm
    flat = torch.randn(5 * 1000 + 7, device="cuda").bfloat16()
    ws = [flat[7 + i * 1000: 7 + i * 1000 + 24 * 16].view(24, 16) for i in range(5)]      # constant stride in one buffer
    out = gemm.stacked_transpose(ws)
    assert out.shape == (5, 16, 24) and all(torch.equal(out[i], ws[i].t()) for i in range(5))
    flat = torch.randn(4 * 400000 + 16, device="cuda").bfloat16()                          # the HIP transpose kernel: dims % 64 == 0
    ws = [flat[16 + i * 400000: 16 + i * 400000 + 384 * 640].view(384, 640) for i in range(4)]
    out = gemm.stacked_transpose(ws)
    assert out.shape == (4, 640, 384) and all(torch.equal(out[i], ws[i].t()) for i in range(4))
    loose = [torch.randn(24, 16, device="cuda").bfloat16() for _ in range(3)]
    out = gemm.stacked_transpose(loose)
    assert all(torch.equal(out[i], loose[i].t()) for i in range(3))


@pytest.mark.parametrize("nb,R,N,K", [(12, 3200, 384, 1536), (4, 8192, 1152, 384), (1, 262144, 256, 128), (3, 96, 128, 128),
                                      (2, 2048, 1536, 384), (1, 102400, 384, 512), (4, 8192, 384, 384)])
def test_gemm_nt_weight_gradient(nb, R, N, K):
    """The NT (weight-gradient) kernel -- LDS-DMA staging, transposed LDS reads, optional deterministic row split -- against an fp32
    torch.mm of the same bf16 operands: fp32 accumulation error only (the products of bf16 numbers are exact in fp32)."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(R + N)
    dy = torch.randn(nb, R, N, device="cuda", generator=g).bfloat16()
    x = torch.randn(nb, R, K, device="cuda", generator=g).bfloat16()
    assert gemm.wgrad_supported(dy, x)
    want = torch.bmm(dy.float().transpose(1, 2), x.float())
    scale = float(want.abs().max())
    for splits in (1, None, 2):
        if splits == 2 and R % 64:
            continue
        got = gemm.wgrad_nt(dy, x, splits=splits)
        assert got.shape == (nb, N, K) and got.dtype == torch.float32
        err = float((got - want).abs().max())
        assert err <= 2e-5 * scale * max(1.0, (R / 4096) ** 0.5), (splits, err, scale)
    # deterministic: two launches agree bit for bit (no atomics anywhere)
    assert torch.equal(gemm.wgrad_nt(dy, x), gemm.wgrad_nt(dy, x))
    # strided operands (a column block of a wider activation) and a destination inside a larger flat buffer
    if N >= 256:
        wide = torch.randn(nb, R, N + 128, device="cuda", generator=g).bfloat16()
        dyv = wide[:, :, 128:]
        flat = torch.zeros(nb * N * K + 64, device="cuda")
        out = flat[64:].view(nb, N, K)
        gemm.wgrad_nt(dyv, x, out)
        want2 = torch.bmm(dyv.float().transpose(1, 2), x.float())
        assert float((out - want2).abs().max()) <= 2e-5 * float(want2.abs().max()) * max(1.0, (R / 4096) ** 0.5)
        assert float(flat[:64].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,N", [(4096, 1536, 384), (3200, 1536, 384), (8192, 1152, 384), (8192, 384, 1152), (200, 64, 128),
                                   (70, 128, 256), (3200, 384, 384)])
@pytest.mark.parametrize("bm", [64, 128])
def test_gemm_ring_equals_register_prefetch_kernel(M, K, N, bm):
    """The LDS-DMA ring form (four stages in LDS, counted vmcnt, source-side swizzle) multiplies in the same order as the
    register-prefetch kernel: results bit-identical, with and without bias, rows past M untouched."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + bm)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    for bias in (None, b):
        want = gemm.linear_tn(x, w, bias)
        out = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        gemm.linear_tn_ring(x, w, bias, out=out[:M], bm=bm)
        assert torch.equal(out[:M], want)
        assert bool((out[M:] == 7.0).all())
    ref = x.float() @ w.float().t()
    assert float((want.float() - (ref + b)).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-2


@pytest.mark.parametrize("M,K,N", [(4096, 384, 1152), (3200, 384, 1536), (8192, 384, 384), (200, 64, 192), (70, 128, 384),
                                   (3328, 1536, 384)])
@pytest.mark.parametrize("bm", [64, 128])
def test_gemm_dma_equals_register_prefetch_kernel(M, K, N, bm):
    """The 192-column LDS-DMA double-buffer form (csrc/gemm_dma.hip) multiplies in the same order as the register-prefetch kernel:
    results bit-identical, with and without bias, rows past M untouched; so is its fc1 (GELU) epilogue."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + bm + 1)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    for bias in (None, b):
        if N % 128 == 0:
            want = gemm.linear_tn(x, w, bias)
        else:       # 192 columns: the register-prefetch kernel has no such tile; fp32 product, one rounding
            want = None
        out = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        gemm.linear_tn_dma(x, w, bias, out=out[:M], bm=bm)
        if want is not None:
            assert torch.equal(out[:M], want)
        ref = x.float() @ w.float().t() + (bias if bias is not None else 0.0)
        assert float((out[:M].float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-2
        assert bool((out[M:] == 7.0).all())
    if N % 128 == 0:
        f0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f1 = torch.empty_like(f0)
        _, g0 = gemm.linear_gelu(x, w, b, f_out=f0)
        _, g1 = gemm.linear_gelu_dma(x, w, b, f_out=f1, bm=bm)
        _, g2 = gemm.linear_gelu_dma(x, w, b, bm=bm)
        assert torch.equal(f0, f1) and torch.equal(g0, g1) and torch.equal(g0, g2)


@pytest.mark.parametrize("M,bm", [(4096, 64), (3200, 64), (8192, 128), (200, 64), (333, 128)])
def test_gemm_dma_gelu_bwd_equals_register_prefetch_kernel(M, bm):
    """fc2 input gradient + GELU backward on csrc/gemm_dma.hip: dF bit-identical to gm3d_gemm_tn_bf16_gelu_bwd; the bias-gradient
    partials are sums of the same fp32 products over other row tiles (another order): equal after the finish within fp32 rounding."""
    from gm3d_amd import gemm
    K, N = 384, 1536
    g = torch.Generator(device="cuda").manual_seed(M + bm)
    d_o = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w2t = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    f = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    df0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cp0 = torch.zeros(gemm.tile_rows(M), N, device="cuda")
    gemm.linear_gelu_bwd(d_o, w2t, f, b, df0, cp0)
    df1 = torch.full((M + 2, N), 3.0, device="cuda", dtype=torch.bfloat16)
    cp1 = torch.zeros((M + bm - 1) // bm, N, device="cuda")
    gemm.linear_gelu_bwd_dma(d_o, w2t, f, b, df1[:M], cp1, bm=bm)
    assert torch.equal(df0, df1[:M]) and bool((df1[M:] == 3.0).all())
    s0, s1 = cp0.sum(0), cp1.sum(0)
    assert float((s0 - s1).abs().max()) <= 1e-5 * float(s0.abs().max()) * max(1.0, M / 4096)


@pytest.mark.parametrize("M,K,N", [(8192, 256, 512), (4100, 512, 256), (8192, 384, 1024), (333, 128, 384), (8192, 384, 128), (200, 64, 768),
                                   (70000, 256, 128)])
@pytest.mark.parametrize("bm", [64, 128])
def test_gemm_dma_tile_widths_equal_register_prefetch_kernel(M, K, N, bm):
    """csrc/gemm_dma.hip with 128- and 256-column tiles (round 3: the mini-PointNet / head products that used to go to the library):
    every width multiplies in the register-prefetch kernel's order -> bit-identical results, with and without bias, rows past M
    untouched; and gemm.mm -- whatever kernel the table names -- returns the same bits."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + N + bm)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    for bias in (None, b):
        want = gemm.linear_tn(x, w, bias) if K // 64 in gemm.OWN_KT else gemm.linear_tn_ring(x, w, bias)
        for bn in (128, 192, 256):
            if N % bn:
                continue
            out = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
            gemm.linear_tn_dmaw(x, w, bias, out=out[:M], bm=bm, bn=bn)
            assert torch.equal(out[:M], want), bn
            assert bool((out[M:] == 7.0).all())
        assert torch.equal(gemm.mm(x, w, bias), want)
        assert gemm.choose(M, N, K) != "lib"
        ref = x.float() @ w.float().t() + (bias if bias is not None else 0.0)
        assert float((want.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-2


@pytest.mark.parametrize("M,K,N", [(8192, 384, 96), (3000, 128, 200), (64, 64, 8), (4096, 384, 136)])
@pytest.mark.parametrize("bm", [64, 128])
def test_gemm_ring_ragged_last_column_tile(M, K, N, bm):
    """The ring kernel with N % 128 != 0 (the 96-wide reconstruction head, P/models_mae_learn_loss.py:169-176): columns [0, N) equal
    the product against the weight zero-padded to whole tiles (same kernel, same order -> same bits), nothing is stored past column
    N or row M."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + N + bm)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    Np = (N + 127) // 128 * 128
    wp = torch.zeros(Np, K, device="cuda", dtype=torch.bfloat16)
    wp[:N] = w
    bp = torch.zeros(Np, device="cuda")
    bp[:N] = b
    for bias, biasp in ((None, None), (b, bp)):
        want = gemm.linear_tn_ring(x, wp, biasp, bm=bm)[:, :N]
        out = torch.full((M + 2, N + 8), 5.0, device="cuda", dtype=torch.bfloat16)
        gemm.linear_tn_ring(x, w, bias, out=out[:M, :N], bm=bm)
        assert torch.equal(out[:M, :N], want)
        assert bool((out[M:] == 5.0).all()) and bool((out[:, N:] == 5.0).all())


def test_gemm_mm_nn_equals_transposed_product():
    """gemm.mm_nn(x, w) = x @ w for a Linear's input gradient: the TN product against a transposed copy of w -- bits equal to
    gemm.mm on an explicitly transposed weight, values equal to the fp32 product within one bf16 rounding."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(5)
    for M, N, K in ((8192, 1024, 384), (102400, 384, 512), (8192, 384, 128), (4096, 256, 128)):
        x = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        w = (torch.randn(N, K, device="cuda", generator=g) / N ** 0.5).bfloat16()
        got = gemm.mm_nn(x, w)
        assert torch.equal(got, gemm.mm(x, w.t().contiguous()))
        ref = x.float() @ w.float()
        assert float((got.float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()) + 1e-2


@pytest.mark.parametrize("M,K,N", [(4096, 1536, 384), (3200, 384, 384), (8192, 1152, 384), (200, 64, 96), (333, 1536, 192), (3200, 128, 1152)])
@pytest.mark.parametrize("bm", [64, 128])
def test_gemm_ring96_equals_ring(M, K, N, bm):
    """The 96-column form of the ring kernel (N = 384 as four column tiles: every CU gets a tile) accumulates along K in the same
    order as the 128-column form: bit-identical results, with and without bias, rows past M untouched."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + N + bm)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    for bias in (None, b):
        want = gemm.linear_tn_ring(x, w, bias, bm=64)
        out = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        gemm.linear_tn_ring96(x, w, bias, out=out[:M], bm=bm)
        assert torch.equal(out[:M], want)
        assert bool((out[M:] == 7.0).all())


@pytest.mark.parametrize("M,K,N", [(65536, 256, 512), (40000, 512, 256), (102400, 384, 512), (70016, 256, 128), (65536, 128, 256),
                                   (102400, 512, 384), (33, 256, 512), (32 * 257 + 5, 512, 384),
                                   # ragged K (96, 288) and narrow N: the 96-wide blocks of the hierarchical encoder's level 0
                                   (65536, 96, 288), (65536, 96, 96), (65536, 96, 384), (65536, 384, 96), (65536, 288, 96),
                                   (40001, 96, 288), (77, 288, 96), (32 * 300 + 9, 96, 384), (262144, 96, 192), (50000, 192, 96), (100000, 96, 512)])
def test_gemm_weight_stationary_equals_tiled_kernels(M, K, N):
    """csrc/gemm_ws.hip (persistent workgroups, W fragments resident in registers, A streamed through an LDS ring by loader waves):
    the same accumulation order along K as the tiled kernels -> bit-identical products, with and without bias, ragged M, nothing
    written past row M."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    for bias in (None, b):
        want = gemm.linear_tn(x, w, bias)
        out = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        gemm.linear_tn_ws(x, w, bias, out=out[:M])
        assert torch.equal(out[:M], want)
        assert bool((out[M:] == 7.0).all())
    if M >= 32768:
        assert torch.equal(gemm.mm(x, w, b), want)


@pytest.mark.parametrize("groups,K,N,after,rows", [(8192, 128, 256, False, True), (3200, 512, 384, True, False), (1030, 128, 256, False, True),
                                                   (1025, 512, 384, True, False), (1100, 512, 384, False, True)])
def test_gemm_weight_stationary_pool_epilogue(groups, K, N, after, rows):
    """conv + max over each group's 32 rows on the weight-stationary kernel == the same product followed by gm3d_group_max_fwd
    (values, argmax decisions and the optional rows, bit for bit)."""
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    M = groups * 32
    g = torch.Generator(device="cuda").manual_seed(groups + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.2
    full = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if rows else None
    pooled = torch.empty(groups, N, device="cuda", dtype=torch.bfloat16)
    arg = torch.empty(groups, N, device="cuda", dtype=torch.uint8)
    check(lib.gm3d_gemm_tn_bf16_ws_pool(_ptr(x), _ptr(w), _ptr(b), _ptr(full), _ptr(pooled), _ptr(arg), M, N, K, K, K, N, N, int(after),
                                        _stream()), "ws_pool")
    z = gemm.linear_tn(x, w, None if after else b)
    if rows:
        assert torch.equal(full, z)
    want = torch.empty(groups, N, device="cuda", dtype=torch.bfloat16)
    warg = torch.empty(groups, N, device="cuda", dtype=torch.uint8)
    check(lib.gm3d_group_max_fwd(_ptr(z), _ptr(b) if after else None, _ptr(want), _ptr(warg), groups, 32, N, 1, _stream()), "gmax")
    assert torch.equal(pooled, want) and torch.equal(arg, warg)


@pytest.mark.parametrize("groups,K,N,after,rows", [(65536, 128, 256, False, True), (65536, 512, 96, True, False), (4099, 128, 256, False, True),
                                                   (2051, 512, 96, True, False), (2049, 512, 96, False, True)])
def test_gemm_weight_stationary_pool_epilogue_groups_of_16(groups, K, N, after, rows):
    """the same epilogue for groups of 16 rows (two groups per 32-row tile: the hierarchical model's level-0 groups of 16 points,
    Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:60) == the product followed by gm3d_group_max_fwd with K = 16, bit for bit,
    including an odd number of groups (the last tile holds one group)."""
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    M = groups * 16
    g = torch.Generator(device="cuda").manual_seed(groups + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    x[16:32] = x[0:16]                       # exact ties between rows of different groups must not leak across the group border
    x[5] = x[3]                              # ... and a tie inside a group goes to the earlier row
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g) * 0.2
    full, pooled, arg = gemm.linear_pool(x, w, b, bias_after_pool=after, want_rows=rows, group_rows=16)
    z = gemm.linear_tn(x, w, None if after else b)
    if rows:
        assert torch.equal(full, z)
    want = torch.empty(groups, N, device="cuda", dtype=torch.bfloat16)
    warg = torch.empty(groups, N, device="cuda", dtype=torch.uint8)
    check(lib.gm3d_group_max_fwd(_ptr(z), _ptr(b) if after else None, _ptr(want), _ptr(warg), groups, 16, N, 1, _stream()), "gmax")
    assert torch.equal(pooled, want) and torch.equal(arg, warg)
    assert int(arg.max()) <= 15


@pytest.mark.parametrize("groups", [4096, 65536, 2062])
def test_gemm_weight_stationary_batchnorm_epilogues_groups_of_16(groups):
    """the same two epilogues with 16 rows per group (two groups per 32-row tile, each lane picks its group's row of T): Point-M2AE's
    level-0 embed (Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:60: 16 points per group)"""
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    M, K, N = groups * 16, 256, 512
    g = torch.Generator(device="cuda").manual_seed(groups)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    t = torch.randn(groups, N, device="cuda", generator=g).bfloat16()
    scale = torch.randn(N, device="cuda", generator=g)
    shift = torch.randn(N, device="cuda", generator=g) * 0.3
    assert gemm.ws_bn_supported(x, w, t, group_rows=16) and not gemm.ws_bn_supported(x, w, t)
    y0 = gemm.linear_tn(x, w)
    want = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    check(lib.gm3d_bn_bcast_apply_relu(_ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(want), groups, 16, N, 0.0, 1, _stream()), "apply")
    got = gemm.linear_ws_bn_apply(x, w, t, scale, shift, group_rows=16)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    prod, part = gemm.linear_ws_bn_stats(x, w, t, group_rows=16)
    assert torch.equal(prod, y0)
    y = (y0.double().view(groups, 16, N) + t.double()[:, None, :]).view(M, N)
    s1, s2 = y.sum(0), (y * y).sum(0)
    st = part.double().sum(0)
    assert float((st[:N] - s1).abs().max()) <= 2e-6 * float(y.abs().sum(0).max())
    assert float((st[N:] - s2).abs().max()) <= 2e-6 * float(s2.max())


@pytest.mark.parametrize("groups", [2048, 8192, 1031])
def test_gemm_weight_stationary_batchnorm_epilogues(groups):
    """second_conv.0 (256 -> 512) with the BatchNorm behind it inside the product's launch (csrc/gemm_ws.hip EPI 4 / 5):
    eval mode  -- act((bf16(x W^T) + t[group]) * scale + shift) == the product followed by gm3d_bn_bcast_apply_relu, bit for bit
                  (negative zeros of slope * h included);
    train mode -- the product itself unchanged, and the per-workgroup partial sums add up to the statistics gm3d_bn_bcast_stats reads
                  the product again for (against an fp64 sum of the same bf16 values; two launches give identical bits)."""
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib, check
    from gm3d_amd.ops import _ptr, _stream
    M, K, N = groups * 32, 256, 512
    g = torch.Generator(device="cuda").manual_seed(groups)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    t = torch.randn(groups, N, device="cuda", generator=g).bfloat16()
    scale = torch.randn(N, device="cuda", generator=g)
    shift = torch.randn(N, device="cuda", generator=g) * 0.3
    assert gemm.ws_bn_supported(x, w, t)
    y0 = gemm.linear_tn(x, w)
    want = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    check(lib.gm3d_bn_bcast_apply_relu(_ptr(y0), _ptr(t), _ptr(scale), _ptr(shift), _ptr(want), groups, 32, N, 0.0, 1, _stream()), "apply")
    got = gemm.linear_ws_bn_apply(x, w, t, scale, shift)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    prod, part = gemm.linear_ws_bn_stats(x, w, t)
    assert torch.equal(prod, y0)
    prod2, part2 = gemm.linear_ws_bn_stats(x, w, t)
    assert torch.equal(part, part2)
    y = (y0.double().view(groups, 32, N) + t.double()[:, None, :]).view(M, N)
    s1, s2 = y.sum(0), (y * y).sum(0)
    st = part.double().sum(0)
    assert float((st[:N] - s1).abs().max()) <= 2e-6 * float(y.abs().sum(0).max())
    assert float((st[N:] - s2).abs().max()) <= 2e-6 * float(s2.max())


@pytest.mark.parametrize("nb,R,N,K", [(4, 8192, 1536, 384), (12, 3328, 384, 1536), (4, 8192, 384, 384), (2, 2048, 1152, 384), (1, 64, 384, 384)])
def test_gemm_nt_big_tiles_equal_small_tiles(nb, R, N, K):
    """128 x 384 output tiles (512-thread workgroups, software-pipelined fragment reads; a measured-slower variant kept behind
    gm3d_gemm_nt_set_big_tiles) against the 128 x 128 kernel: the same 32-row stages in the same order -> bit-identical slabs for equal
    row splits, strided operands included."""
    from gm3d_amd import gemm
    from gm3d_amd._capi import lib
    g = torch.Generator(device="cuda").manual_seed(R + N + K)
    wide = torch.randn(nb, R, N + 64, device="cuda", generator=g).bfloat16()
    dy = wide[:, :, 64:]
    x = torch.randn(nb, R, K, device="cuda", generator=g).bfloat16()
    try:
        for splits in (1, 2, 4):
            if R % (32 * splits):
                continue
            lib.gm3d_gemm_nt_set_big_tiles(0)
            small = gemm.wgrad_nt(dy, x, splits=splits)
            lib.gm3d_gemm_nt_set_big_tiles(1)
            big = gemm.wgrad_nt(dy, x, splits=splits)
            assert torch.equal(big, small), splits
    finally:
        lib.gm3d_gemm_nt_set_big_tiles(0)


@pytest.mark.parametrize("M,K,N", [(65536, 96, 288), (8192, 96, 96), (3000, 288, 96), (4096, 96, 384), (200, 40, 72), (8192, 192, 576), (333, 8, 8),
                                   (5000, 576, 192), (64, 96, 128)])
def test_gemm_ragged_shapes_on_the_register_prefetch_kernel(M, K, N):
    """csrc/gemm.hip's ragged form (N, K multiples of 8: Point-M2AE's 96 / 192 / 288 / 576-wide layers): W rows past N clamped, the last
    K-stage zero-filled, column chunks past N not stored -- against an fp32 product of the same bf16 operands (fp32 accumulation, one
    rounding), with and without bias, nothing written outside the (M,N) block of a wider destination; gemm.mm / mm_nn route here."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    assert gemm.ragged_supported(x, w)
    for bias in (None, b):
        want = x.float() @ w.float().t() + (bias if bias is not None else 0.0)
        wide = torch.full((M + 2, N + 8), 7.0, device="cuda", dtype=torch.bfloat16)
        gemm.linear_tn(x, w, bias, out=wide[:M, :N])
        err = float((wide[:M, :N].float() - want).abs().max())
        assert err <= 1e-2 * max(1.0, float(want.abs().max())), err
        assert bool((wide[M:] == 7.0).all()) and bool((wide[:, N:] == 7.0).all())
        got = gemm.mm(x, w, bias)
        assert float((got.float() - want).abs().max()) <= 1e-2 * max(1.0, float(want.abs().max()))
    wt = w.t().contiguous()                     # (K,N): mm_nn(x, wt) = x @ wt
    got = gemm.mm_nn(x, wt)
    want = x.float() @ wt.float()
    assert float((got.float() - want).abs().max()) <= 1e-2 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("nb,R,N,K", [(1, 65536, 288, 96), (3, 2048, 96, 96), (1, 8192, 96, 384), (2, 4096, 576, 192), (1, 64, 8, 8), (1, 32768, 192, 768)])
def test_gemm_nt_ragged_edges(nb, R, N, K):
    """The weight-gradient kernel with N, K any multiples of 8 (edge tiles read clamped chunks and store only what exists) against an
    fp32 product; the rows of a wider destination beyond (N,K) stay untouched; deterministic."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(R + N + K)
    dy = torch.randn(nb, R, N, device="cuda", generator=g).bfloat16()
    x = torch.randn(nb, R, K, device="cuda", generator=g).bfloat16()
    assert gemm.wgrad_supported(dy, x)
    want = torch.bmm(dy.float().transpose(1, 2), x.float())
    got = gemm.wgrad_nt(dy, x)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) * max(1.0, (R / 4096) ** 0.5)
    assert torch.equal(got, gemm.wgrad_nt(dy, x))
    flat = torch.full((nb * N * K + 256,), 3.0, device="cuda")
    out = flat[128:128 + nb * N * K].view(nb, N, K)
    gemm.wgrad_nt(dy, x, out)
    assert torch.equal(out, got) and bool((flat[:128] == 3.0).all()) and bool((flat[128 + nb * N * K:] == 3.0).all())


@pytest.mark.parametrize("nb,R,N,K,splits", [(4, 8192, 1536, 384, 4), (4, 8192, 384, 384, 8), (12, 3328, 1152, 384, 2), (1, 8192, 96, 288, 4),
                                              (2, 2048, 384, 1536, 2), (1, 4096, 576, 192, 8)])
def test_gemm_nt_slab_sum_inside_the_launch(nb, R, N, K, splits):
    """gm3d_gemm_nt_bf16_sum (the workgroup that finishes a tile's last slab adds the tile's slabs in slab order) == the slabs followed by
    gm3d_sum_few_rows, bit for bit, on 20 consecutive launches (the arrival order of the workgroups varies, the result must not);
    the tile counters are zero again afterwards; a pitched destination inside a larger buffer is respected.  (Off in the step:
    measured slower, gemm.FUSE_SLAB_SUM.)"""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(R + N + K + splits)
    dy = torch.randn(nb, R, N, device="cuda", generator=g).bfloat16()
    x = torch.randn(nb, R, K, device="cuda", generator=g).bfloat16()
    was = gemm.FUSE_SLAB_SUM
    try:
        gemm.FUSE_SLAB_SUM = False
        want = gemm.wgrad_nt(dy, x, splits=splits)
        gemm.FUSE_SLAB_SUM = True
        for _ in range(20):
            got = gemm.wgrad_nt(dy, x, splits=splits)
            assert torch.equal(got, want)
        buf = gemm._nt_counters[str(dy.device)][0]
        assert int(buf.abs().max()) == 0
        flat = torch.full((nb * N * K + 512,), 3.0, device="cuda")
        out = flat[256:256 + nb * N * K].view(nb, N, K)
        gemm.wgrad_nt(dy, x, out, splits=splits)
        assert torch.equal(out, want) and bool((flat[:256] == 3.0).all()) and bool((flat[256 + nb * N * K:] == 3.0).all())
    finally:
        gemm.FUSE_SLAB_SUM = was


def test_gemm_nt_multi_problem_launch_equals_separate_launches():
    """gm3d_gemm_nt_bf16_multi: the twelve weight-gradient products of three block stacks (decoders: 4 blocks x 8192 rows, encoder: 12 x
    3328) as ONE launch + ONE slab-sum launch == wgrad_nt per request, bit for bit; destinations at a batch stride inside a larger flat
    buffer (the optimizer's gradient slots) stay clean around the slots."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(4)
    reqs, flats = [], []
    for nb, R in ((4, 8192), (4, 8192), (12, 3328)):
        for N, K in ((384, 1536), (1536, 384), (384, 384), (1152, 384)):
            dy = torch.randn(nb, R, N, device="cuda", generator=g).bfloat16()
            x = torch.randn(nb, R, K, device="cuda", generator=g).bfloat16()
            stride = N * K + 256                                     # slots of a flat buffer with other parameters in between
            flat = torch.full((nb * stride + 64,), 3.0, device="cuda")
            out = torch.as_strided(flat, (nb, N, K), (stride, K, 1), 64)
            reqs.append((dy, x, out))
            flats.append((flat, stride, N * K, nb))
    assert all(gemm.wgrad_multi_ok(*r) for r in reqs)
    got, used = gemm.wgrad_nt_multi(reqs, want_splits=True)             # the launch's own (capped) row splits
    assert max(used) <= max(gemm.MULTI_SPLITS_MAX, 1) or not gemm.MULTI_SPLITS_MAX
    want = [gemm.wgrad_nt(dy, x, splits=sp) for (dy, x, _), sp in zip(reqs, used)]
    for w, o, (flat, stride, nk, nb) in zip(want, got, flats):
        assert torch.equal(o, w)
        assert bool((flat[:64] == 3.0).all())
        for b in range(nb):
            assert bool((flat[64 + b * stride + nk:64 + (b + 1) * stride] == 3.0).all())
    own = [gemm.lib.gm3d_gemm_nt_splits(dy.shape[0], dy.shape[1], dy.shape[2], x.shape[2]) for dy, x, _ in reqs[:5]]
    got2 = gemm.wgrad_nt_multi([(dy, x, None) for dy, x, _ in reqs[:5]], splits=own)      # given splits, fresh destinations
    for (dy, x, _), o in zip(reqs[:5], got2):
        assert torch.equal(o, gemm.wgrad_nt(dy, x))


def test_stacked_transposes_one_launch_equals_one_per_group():
    """gemm.stacked_transposes (gm3d_transpose_bf16_multi: the four transposed weight shadows of a stack in one launch) == one
    stacked_transpose per group, for weights at a constant stride inside one flat buffer (the optimizer's bf16 shadow)."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(2)
    nblk = 4
    shapes = [(384, 1536), (384, 384), (1536, 384), (1152, 384)]
    per = sum(a * b for a, b in shapes) + 640
    flat = torch.randn(nblk * per, device="cuda", generator=g).bfloat16()
    groups, off = [], 0
    for a, b in shapes:
        groups.append([flat[i * per + off:i * per + off + a * b].view(a, b) for i in range(nblk)])
        off += a * b
    got = gemm.stacked_transposes(groups)
    for ws, o in zip(groups, got):
        assert torch.equal(o, torch.stack(ws).transpose(1, 2).contiguous())
    one = gemm.stacked_transposes([groups[0]])            # a single group: the per-group path
    assert torch.equal(one[0], got[0])


def test_wgrad_multi_default_splits_depend_on_the_batch_only_in_the_last_bits():
    """wgrad_nt_multi picks a request's row splits from the whole launch's tile count, so the fp32 summation order of one
    weight gradient depends on what it is batched with.  With explicit splits the result is wgrad_nt's to the bit; with the defaults the
    same product alone / in a large launch agrees to fp32 rounding of the sum (1e-6 of the largest entry), nothing more."""
    from gm3d_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda nb, R, N, K: ((torch.randn(nb, R, N, device="cuda", generator=g) * 0.1).bfloat16(), torch.randn(nb, R, K, device="cuda", generator=g).bfloat16())
    small = mk(4, 8192, 384, 384)
    others = [mk(12, 8192, 1536, 384), mk(12, 8192, 384, 1536), mk(12, 8192, 1152, 384)]
    alone, spl_a = gemm.wgrad_nt_multi([(small[0], small[1], None), (others[0][0][:1], others[0][1][:1], None)], want_splits=True)
    batched, spl_b = gemm.wgrad_nt_multi([(small[0], small[1], None)] + [(a, b, None) for a, b in others], want_splits=True)
    ref = gemm.wgrad_nt(small[0], small[1], splits=spl_b[0])
    assert torch.equal(batched[0], ref)                                   # explicit splits == the per-request kernel, bit for bit
    assert torch.equal(alone[0], gemm.wgrad_nt(small[0], small[1], splits=spl_a[0]))
    scale = float(ref.abs().max())
    assert float((alone[0] - batched[0]).abs().max()) <= 2e-6 * scale
