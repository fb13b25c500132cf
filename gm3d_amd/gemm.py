"""bf16 GEMM of the block / mini-PointNet shapes on the hand-written MFMA kernel (csrc/gemm.hip): y = x @ w^T (+ bias), and
its fused-epilogue forms (fc1+GELU, fc2-dgrad+GELU', conv+max-pool).  DESIGN.md 3b' has the design and the measurements;
`choose` names the kernel for every plain shape (table: tools/gemm_kbench.py -> profiles/r02_gemm_kbench.txt)."""
import torch

from ._capi import lib
from .ops import _launch, _ptr, _stream

import os

ENABLED = True            # module attributes below: flipped by the equality tests and tools/, not environment switches
FUSE_GELU = True          # fc1 + bias + GELU as one launch in the fused transformer stack
FUSE_POOL = True          # mini-PointNet conv + max-pool as one launch
FUSE_GELU_BWD = True      # fc2 input gradient + GELU backward + fc1 bias-gradient partials


def supported(x, w):
    """x (M,K) bf16 with unit inner stride, w (N,K) bf16 contiguous rows; N % 128 == 0, K % 64 == 0."""
    return (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
            and x.shape[1] == w.shape[1] and w.shape[0] % 128 == 0 and w.shape[1] % 64 == 0 and x.stride(1) == 1
            and w.stride(1) == 1 and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and x.stride(0) >= x.shape[1]
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


RAGGED = True            # widths that are multiples of 8 but not of the tiles (Point-M2AE's 96 / 192 / 288 / 576) on our own kernels


def ragged_supported(x, w):
    """the plain product on csrc/gemm.hip's ragged form: N, K any multiples of 8 (W rows past N clamped, the last K-stage zero-filled)."""
    return (ENABLED and RAGGED and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
            and x.shape[1] == w.shape[1] and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0 and w.shape[1] >= 8 and x.stride(1) == 1
            and w.stride(1) == 1 and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and x.stride(0) >= x.shape[1]
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def linear_tn(x, w, bias=None, out=None):
    """x (M,K) bf16, w (N,K) bf16, bias (N) f32 or None -> (M,N) bf16 = x @ w^T + bias (fp32 accumulate, one rounding)."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N, K,
            x.stride(0), w.stride(0), out.stride(0), _stream())
    return out


def linear_tn_ring(x, w, bias=None, out=None, bm=None):
    """linear_tn through the LDS-DMA ring kernel (csrc/gemm_ring.hip): same results bit for bit; for long K over few tiles."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    if bm is None:
        bm = 64 if M <= 4096 else 128
    _launch("gm3d_gemm_tn_bf16_ring", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ring, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N,
            K, x.stride(0), w.stride(0), out.stride(0), int(bm), _stream())
    return out


def linear_tn_ring96(x, w, bias=None, out=None, bm=64):
    """linear_tn through the ring kernel with 96-column tiles (N % 96 == 0; csrc/gemm_ring.hip): bit-identical."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_ring96", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ring96, _ptr(x), _ptr(w), _ptr(bias), _ptr(out),
            M, N, K, x.stride(0), w.stride(0), out.stride(0), int(bm), _stream())
    return out


USE_DMA = True            # K = 384 products over >= 6 column tiles of 192 on csrc/gemm_dma.hip (qkv, fc1 + GELU, fc2 input gradient)


def dma_bm(M):
    """Tile height of the 192-column kernel: 64 rows up to 4096 (where 128-row tiles leave CUs without a tile), else 128."""
    return 64 if M <= 4096 else 128


def dma_supported(x, w):
    return USE_DMA and supported(x, w) and w.shape[0] % 192 == 0 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0


def linear_tn_dma(x, w, bias=None, out=None, bm=64):
    """linear_tn through the 192-column LDS-DMA double-buffer kernel (csrc/gemm_dma.hip): same results bit for bit; for K = 384."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_dma", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_dma, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N,
            K, x.stride(0), w.stride(0), out.stride(0), int(bm), _stream())
    return out


def linear_gelu_dma(x, w, bias, f_out=None, g_out=None, bm=64):
    """linear_gelu on csrc/gemm_dma.hip (bit-identical)."""
    M, K = x.shape
    N = w.shape[0]
    if g_out is None:
        g_out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_dma_gelu", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_dma_gelu, _ptr(x), _ptr(w), _ptr(bias),
            _ptr(f_out), _ptr(g_out), M, N, K, x.stride(0), w.stride(0), f_out.stride(0) if f_out is not None else 0,
            g_out.stride(0), int(bm), _stream())
    return f_out, g_out


# LayerNorm folded into the producer / consumer GEMMs (fused.py).  OFF: measured on MI355X at B = 128 (same-box A/B, round 2) the
# step is 2.5 % SLOWER with it (8.50 vs 8.29 ms) although 97 of 111 LayerNorm launches disappear -- the pass is dominated by the
# fp32 residual stream (read 4 + write 4 of its 14 bytes per element), which the fusion cannot remove, only move into GEMM
# epilogues where it is serialised behind the MFMA loop (+5.7 us per producer, +6 us per consumer vs 6.5-10 us per LayerNorm
# launch).  Kept as tested kernels (tests/test_gpu_fused_ln.py); not an environment switch.
FUSE_LN = False


def linear_res(x, w, bias, res, rowscale, rows_per_sample, add, bm=None):
    """proj / fc2 with the residual epilogue: -> (U (M,384) f32 = res + rowscale * (bf16(x @ w^T) + bias) + add, its bf16 copy,
    stats (3,M,2) f32 = per 128-column tile (mean, sum of squared deviations))."""
    M, K = x.shape
    N = w.shape[0]
    U = torch.empty(M, N, dtype=torch.float32, device=x.device)
    U16 = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    stats = torch.empty(3, M, 2, dtype=torch.float32, device=x.device)
    if bm is None:
        bm = 64 if M <= 4096 else 128
    _launch("gm3d_gemm_tn_bf16_res", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_res, _ptr(x), _ptr(w), _ptr(bias), _ptr(res),
            _ptr(rowscale), int(rows_per_sample), _ptr(add), _ptr(U), _ptr(U16), _ptr(stats), M, N, K, x.stride(0), w.stride(0), int(bm),
            _stream())
    return U, U16, stats


def linear_lna(U, stats, gamma, beta, eps, w, bias=None, gelu=False, f_out=None, g_out=None, h_out=None, want_stats=False):
    """LayerNorm(U) @ w^T (+ bias) with the normalisation applied while the A operand is staged; U = the bf16 copy of the stream.
    gelu=False -> (C (M,N) bf16, mean, rstd); gelu=True -> (f_out | None, G = GELU(.. + bias), mean, rstd).  h_out (M,384) bf16
    receives the normalised rows, mean / rstd (M) f32 are produced when want_stats."""
    M, K = U.shape
    N = w.shape[0]
    dev = U.device
    mean = torch.empty(M, dtype=torch.float32, device=dev) if want_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=dev) if want_stats else None
    if gelu:
        if g_out is None:
            g_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        _launch("gm3d_gemm_tn_bf16_lna", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_lna, _ptr(U), _ptr(stats), _ptr(gamma),
                _ptr(beta), float(eps), _ptr(w), _ptr(bias), _ptr(f_out), _ptr(g_out), _ptr(h_out), _ptr(mean), _ptr(rstd), M, N, K,
                U.stride(0), w.stride(0), f_out.stride(0) if f_out is not None else 0, g_out.stride(0), _stream())
        return f_out, g_out, mean, rstd
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    _launch("gm3d_gemm_tn_bf16_lna", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_lna, _ptr(U), _ptr(stats), _ptr(gamma), _ptr(beta),
            float(eps), _ptr(w), _ptr(bias), _ptr(c), None, _ptr(h_out), _ptr(mean), _ptr(rstd), M, N, K, U.stride(0), w.stride(0),
            c.stride(0), 0, _stream())
    return c, mean, rstd


def linear_gelu(x, w, bias, f_out=None, g_out=None):
    """fc1 -> GELU with the activation in the GEMM epilogue: g = GELU(x @ w^T + bias); `f_out` (optional) receives the
    bf16 pre-activation WITHOUT bias (what bias_gelu_bwd re-reads).  -> (f_out | None, g)."""
    M, K = x.shape
    N = w.shape[0]
    if g_out is None:
        g_out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_gelu", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_gelu, _ptr(x), _ptr(w), _ptr(bias),
            _ptr(f_out), _ptr(g_out), M, N, K, x.stride(0), w.stride(0), f_out.stride(0) if f_out is not None else 0,
            g_out.stride(0), _stream())
    return f_out, g_out


USE_WS = True             # tall products (M >= 32768) of the mini-PointNet shapes on the weight-stationary kernel (csrc/gemm_ws.hip):
#                           40 / 104 / 55 / 104 / 43 us against 53 / 142 / 76 / 112 / 47 on the tiled kernels (r03 table), -0.13 ms per step
USE_WS_POOL = True        # ... and conv + max-pool (the product computed non-transposed for these tiles: lane = column, so the pool is a
#                           register max + one cross-lane step): 80 / 138 / 55 us against 80 / 144 / 57 on csrc/gemm_dma.hip, +0.4 % in the
#                           step (same-box A/B 7.857 vs 7.890 ms).  (A first form -- DPP reduction over the rows of the transposed tile, ~700
#                           VALU instructions per wave and tile -- took 102 / 218 us.)


def ws_supported(x, w, pool=False):
    return ((USE_WS_POOL if pool else USE_WS) and supported(x, w) and x.shape[0] >= 32768
            and bool(lib.gm3d_gemm_ws_supported(w.shape[0], w.shape[1], int(pool))))


DMA192_BM64 = True       # ... with 64-row tiles while 128-row tiles would not give every CU two workgroups (one column tile at N = 192)
DMA192_RAGGED = True     # N % 192 == 0 but N % 128 != 0 (192, 576) at K % 64 == 0: 192-column double-buffer tiles (A/B: tools/ab_m2ae.sh)
WS_RAGGED = True         # ... and the ragged-K members (K = 96 / 288; N = 96 / 288 / 384): 25-28 us on the tiled kernels at 65,536 rows


def ws_ragged_supported(x, w):
    return (USE_WS and WS_RAGGED and ragged_supported(x, w) and x.shape[0] >= 32768
            and (w.shape[0] % 128 != 0 or w.shape[1] % 64 != 0) and bool(lib.gm3d_gemm_ws_supported(w.shape[0], w.shape[1], 0)))


def linear_tn_ws(x, w, bias=None, out=None):
    """linear_tn through the weight-stationary streaming kernel (csrc/gemm_ws.hip): bit-identical; for the tall mini-PointNet products."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_ws", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ws, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), M, N, K,
            x.stride(0), w.stride(0), out.stride(0), _stream())
    return out


WS_BN = True             # second_conv.0 with the BatchNorm behind it in the product's epilogue (eval: apply + ReLU; train: the statistics)


WS_BN_STATS16 = False    # the train-mode statistics epilogue for 16-row groups: slower than the separate pass (see embed.py); kept as a tested entry point


def ws_bn_supported(x, w, t, group_rows=32):
    return (WS_BN and group_rows in (16, 32) and ws_supported(x, w) and x.shape[0] % 32 == 0 and t.dtype == torch.bfloat16 and t.is_contiguous()
            and t.shape == (x.shape[0] // group_rows, w.shape[0]) and lib.gm3d_gemm_ws_stats_rows(x.shape[0], w.shape[0], x.shape[1]) > 0)


def linear_ws_bn_apply(x, w, t, scale, shift, slope=0.0, group_rows=32):
    """act((bf16(x @ w^T) + t[row // 32]) * scale + shift) in ONE launch (eval-mode BatchNorm + ReLU in the epilogue of csrc/gemm_ws.hip);
    bit-identical to linear_tn_ws followed by gm3d_bn_bcast_apply_relu."""
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_ws_bn_apply", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ws_bn_apply_g, _ptr(x), _ptr(w), _ptr(t),
            _ptr(scale), _ptr(shift), float(slope), _ptr(out), M, N, K, x.stride(0), w.stride(0), t.stride(0), out.stride(0), int(group_rows),
            _stream())
    return out


def linear_ws_bn_stats(x, w, t, group_rows=32):
    """-> (bf16(x @ w^T), partial (rows, 2 N) f32): the product and, per workgroup, the column sums of y = product + t[row // 32] and
    y^2 (the train-mode statistics of the BatchNorm behind it; sum the rows in order)."""
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    rows = lib.gm3d_gemm_ws_stats_rows(M, N, K)
    part = torch.empty(rows, 2 * N, dtype=torch.float32, device=x.device)
    _launch("gm3d_gemm_tn_bf16_ws_bn_stats", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ws_bn_stats_g, _ptr(x), _ptr(w), _ptr(t), _ptr(out),
            _ptr(part), M, N, K, x.stride(0), w.stride(0), t.stride(0), out.stride(0), int(group_rows), _stream())
    return out, part


def linear_tn_dmaw(x, w, bias=None, out=None, bm=128, bn=256):
    """linear_tn through csrc/gemm_dma.hip with a bm x bn tile (bn = 128 / 192 / 256 columns, N % bn == 0): bit-identical."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    _launch("gm3d_gemm_tn_bf16_dmaw", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_dmaw, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), M,
            N, K, x.stride(0), w.stride(0), out.stride(0), int(bm), int(bn), _stream())
    return out


N384_BIG = "ring128"      # N = 384 above 4096 rows.  "dma64x128" (two workgroups per CU) is 2-8 % faster launch by launch
#                           (profiles/r03_gemm_kbench.txt) and 1.1 % SLOWER in the step (same-box A/B, gpurun_out/r3i: 8.08 vs 7.99 ms):
#                           these launches run beside the other decoder's chain, which the second workgroup per CU displaces
OWN_KT = (2, 4, 6, 8, 16, 18, 24)      # K / 64 values csrc/gemm.hip is unrolled for


def choose(M, N, K):
    """Which hand-written kernel runs the plain product x (M,K) @ w (N,K)^T on MI355X: "own" (register-prefetch kernel,
    csrc/gemm.hip), "ring64" / "ring128" (LDS-DMA ring, csrc/gemm_ring.hip, tile height), "dma<bm>x<bn>" (LDS-DMA double buffer
    with a bm x bn tile, csrc/gemm_dma.hip).  From the per-shape tables profiles/r03_gemm_kbench.txt (tools/gemm_kbench.py).
    Round 3: NO shape goes to the library any more (round 2 kept a shape there where the tuned hipBLASLt solution was >= 5 %
    faster): every product of the bf16 step is one of our kernels with a fixed accumulation order -- eager == captured by
    construction, and the measured path no longer depends on a TunableOp table."""
    if N == 384 and K in (384, 1024, 1152, 1536):       # three column tiles: 150 .. 192 workgroups for 256 CUs
        # the ring, one workgroup per CU, two K-stages in flight.  Neither 96-column tiles (every CU busy, fewer bytes per
        # workgroup) nor a deeper ring helps: the r03 table has both slower -- the chip-wide L2 -> LDS rate is what is exhausted
        return "ring64" if M <= 4096 else N384_BIG
    if (N, K) == (1152, 384):
        return "ring128" if M <= 3200 else ("dma64x192" if M <= 4096 else "dma128x192") if USE_DMA else "own"
    if K == 384 and N % 192 == 0 and N > 384 and M < 32768:
        return ("dma64x192" if M <= 4096 else "dma128x192") if USE_DMA else "own"
    if M >= 32768:                                       # the mini-PointNet products over the point rows: HBM-bound streams
        # two workgroups per CU with one 64-KiB stage each in flight beat every deeper / wider form measured (r03 table:
        # 128x128 142 us vs 128x256 145, ring128 174, register prefetch 174 at 262144 x 256 -> 512)
        return "dma128x192" if N % 192 == 0 else "dma128x128"
    if (N, K) == (256, 512) or (N, K) == (128, 384):    # per-group / per-token input gradients: few tiles, longer K
        return "ring64"
    if N == 1024:
        return "dma128x128"
    return "dma64x128"


def mm(x, w, bias=None, out=None):
    """x (M,K) @ w (N,K)^T (+ bias) on the kernel `choose` names.  Operands that do not meet the hand-written kernels' layout
    rules (fp32 parity mode, odd widths) go to torch.mm: never on the bf16 step's path (tools/leftover_sites.py lists none)."""
    how = choose(x.shape[0], w.shape[0], w.shape[1]) if supported(x, w) else "lib"
    if how == "lib" and ws_ragged_supported(x, w) and (out is None or (out.stride(1) == 1 and out.stride(0) % 8 == 0)):
        return linear_tn_ws(x, w, bias, out)       # the 96-wide blocks over 65,536 token rows (Point-M2AE level 0): HBM streams
    if (how == "lib" and DMA192_RAGGED and USE_DMA and ragged_supported(x, w) and w.shape[0] % 192 == 0 and w.shape[1] % 64 == 0
            and x.shape[0] >= 8192 and (out is None or (out.stride(1) == 1 and out.stride(0) % 8 == 0))):
        # N = 192 / 576 (the 192-wide level of the hierarchical model): whole 192-column tiles on the LDS-DMA double-buffer kernel instead of
        # the ring kernel's 128-column tiles with a ragged second one (a quarter of its MFMA and LDS work wasted)
        tiles128 = -(-x.shape[0] // 128) * (w.shape[0] // 192)
        return linear_tn_dmaw(x, w, bias, out, bm=64 if (DMA192_BM64 and tiles128 < 512) else dma_bm(x.shape[0]), bn=192)
    if how == "lib" and ragged_supported(x, w) and (out is None or (out.stride(1) == 1 and out.stride(0) % 8 == 0)):
        # a ragged last column tile alone (K a multiple of the 64-column stage): the ring kernel; a ragged K as well: csrc/gemm.hip
        if w.shape[1] % 64 == 0:
            return linear_tn_ring(x, w, bias, out)
        return linear_tn(x, w, bias, out)
    if how != "lib" and ws_supported(x, w):
        return linear_tn_ws(x, w, bias, out)
    if how == "own":
        return linear_tn(x, w, bias, out)
    if how.startswith("ring"):
        return linear_tn_ring(x, w, bias, out, bm=int(how[4:]))
    if how.startswith("dma"):
        bm, bn = how[3:].split("x")
        return linear_tn_dmaw(x, w, bias, out, bm=int(bm), bn=int(bn))
    y = x @ w.t()
    if bias is not None:
        y = y + bias.to(y.dtype)
    return y if out is None else out.copy_(y)


def transposed(w):
    """w (N,K) bf16 contiguous -> (K,N) contiguous, one launch of our transpose kernel (N, K % 8 == 0), else torch."""
    N, K = w.shape
    if w.dtype == torch.bfloat16 and w.is_contiguous() and N % 8 == 0 and K % 8 == 0 and w.is_cuda and w.data_ptr() % 16 == 0:
        out = torch.empty(K, N, dtype=w.dtype, device=w.device)
        _launch("gm3d_transpose_bf16_batched", {"batch": 1, "rows": N, "cols": K}, lib.gm3d_transpose_bf16_batched, _ptr(w), _ptr(out),
                1, N, K, N * K, _stream())
        return out
    return w.t().contiguous()


# Transposed shadows of the weights mm_nn serves, made in ONE launch per eight weights at the start of a backward region instead of one
# launch per layer inside it (Point-M2AE: 19 small layers).  mm_nn remembers the bf16 shadows it was asked about (views of the flat
# optimizer's shadow buffer: same address every step; forgotten whenever an optimizer registers new shadows -- forget_transposes);
# prepare_transposes() transposes them; the cache holds for the region only (the weights change with every optimizer step).
_tr_known = {}
_tr_cache = None


def _tr_key(w):
    return (w.data_ptr(), tuple(w.shape))


def forget_transposes():
    _tr_known.clear()
    drop_transposes()


def prepare_transposes():
    global _tr_cache
    live = list(_tr_known.values())
    _tr_cache = {}
    for i in range(0, len(live), 8):
        chunk = live[i:i + 8]
        outs = stacked_transposes([[w] for w in chunk]) if len(chunk) > 1 else [transposed(chunk[0]).unsqueeze(0)]
        for w, o in zip(chunk, outs):
            _tr_cache[_tr_key(w)] = o[0]


def drop_transposes():
    global _tr_cache
    _tr_cache = None


def _transposed_for_mm_nn(w):
    if _tr_cache is not None:
        hit = _tr_cache.get(_tr_key(w))
        if hit is not None:
            return hit
        if (len(_tr_known) < 64 and w.is_contiguous() and w.dtype == torch.bfloat16 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
                and w.data_ptr() % 16 == 0 and not torch.cuda.is_current_stream_capturing()):
            _tr_known.setdefault(_tr_key(w), w)           # asked for inside a region: ready at the start of the next one
    return transposed(w.contiguous())


def mm_nn(x, w, out=None):
    """x (M,N) @ w (N,K) -> (M,K): the input gradient dY . W of a Linear / Conv1d(k=1) layer with weight w (N,K).  bf16 operands
    run as the TN product against a transposed copy of w (one small transposing launch per call: the weights of the layers this
    serves have 32 K .. 512 K elements); anything else goes to torch.mm."""
    if (ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
            and w.shape[1] % 8 == 0 and w.shape[0] % 8 == 0 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0):
        wt = _transposed_for_mm_nn(w)
        if supported(x, wt) or ragged_supported(x, wt):
            return mm(x, wt, None, out)
    y = x @ w
    return y if out is None else out.copy_(y)


POOL_ON_DMA = True        # the pool epilogue on csrc/gemm_dma.hip (128 x 128 / 128 x 192 tiles) instead of csrc/gemm.hip


def pool16_supported(x, w):
    """conv + max over groups of 16 rows in the weight-stationary kernel's epilogue (the hierarchical model's level-0 groups)"""
    N, K = w.shape
    return (USE_WS_POOL and ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2
            and x.shape[0] >= 32768 and x.shape[0] % 16 == 0 and x.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1
            and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and bool(lib.gm3d_gemm_ws_supported(N, K, 1)))


def linear_pool(x, w, bias, bias_after_pool, want_rows, group_rows=32):
    """Conv1d(k=1) over (groups*group_rows, K) rows + max over each group's rows in the GEMM epilogue (group_rows 32; 16 on the
    weight-stationary kernel only: pool16_supported).
    -> (rows (M,N) bf16 | None, pooled (M/group_rows,N) bf16, argmax (M/group_rows,N) uint8)."""
    M, K = x.shape
    N = w.shape[0]
    rows = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if want_rows else None
    pooled = torch.empty(M // group_rows, N, dtype=torch.bfloat16, device=x.device)
    arg = torch.empty(M // group_rows, N, dtype=torch.uint8, device=x.device)
    if group_rows == 16:
        _launch("gm3d_gemm_tn_bf16_ws_pool", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ws_poolg, _ptr(x), _ptr(w), _ptr(bias), _ptr(rows),
                _ptr(pooled), _ptr(arg), M, N, K, x.stride(0), w.stride(0), N, N, int(bias_after_pool), 16, _stream())
        return rows, pooled, arg
    if ws_supported(x, w, pool=True):
        _launch("gm3d_gemm_tn_bf16_ws_pool", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_ws_pool, _ptr(x), _ptr(w), _ptr(bias), _ptr(rows),
                _ptr(pooled), _ptr(arg), M, N, K, x.stride(0), w.stride(0), N, N, int(bias_after_pool), _stream())
        return rows, pooled, arg
    if POOL_ON_DMA and N % 8 == 0 and (N % 192 == 0 or N % 128 == 0):
        bn = 192 if N % 192 == 0 else 128
        _launch("gm3d_gemm_tn_bf16_dma_pool", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_dma_pool, _ptr(x), _ptr(w), _ptr(bias),
                _ptr(rows), _ptr(pooled), _ptr(arg), M, N, K, x.stride(0), w.stride(0), N, N, int(bias_after_pool), 128, bn, _stream())
        return rows, pooled, arg
    _launch("gm3d_gemm_tn_bf16_pool", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_pool, _ptr(x), _ptr(w), _ptr(bias), _ptr(rows),
            _ptr(pooled), _ptr(arg), M, N, K, x.stride(0), w.stride(0), N, N, int(bias_after_pool), _stream())
    return rows, pooled, arg


OWN_WGRAD = True          # weight-gradient (NT) GEMMs on the hand-written kernel (csrc/gemm_nt.hip)


def wgrad_supported(dy, x):
    """dy (nb,R,N), x (nb,R,K) bf16 with unit inner stride and a common batch layout; N, K % 8 == 0 (ragged edge tiles), R % 32 == 0."""
    edge = 8 if RAGGED else 128
    return (ENABLED and OWN_WGRAD and dy.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.dim() == 3
            and x.dim() == 3 and dy.shape[:2] == x.shape[:2] and dy.shape[2] % edge == 0 and x.shape[2] % edge == 0
            and dy.shape[1] % 32 == 0 and dy.stride(2) == 1 and x.stride(2) == 1 and dy.stride(1) % 8 == 0 and x.stride(1) % 8 == 0
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and (dy.shape[0] == 1 or (dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0)))


WGRAD_SPLITS_CAP = 0        # measurement knob: cap on the row splits of the WGRAD_SPLITS_CAP_NB-block stacks' products (0 = none)
WGRAD_SPLITS_CAP_NB = 4


FUSE_SLAB_SUM = False      # the sum over the row splits inside the NT launch (the last workgroup of a tile adds its slabs in slab order):
#                            bit-identical, 17 launches fewer per step -- and 0.8-1.0 ms SLOWER (same-box A/B 8.32 / 8.59 vs 7.53 ms): the
#                            agent-scope release fence every workgroup needs before it bumps the tile counter writes back its XCD's L2
#                            (8 XCDs, no common L2), also under the kernels running beside it.  Kept as a tested entry point.
FUSE_SLAB_SUM_MAX = 8      # ... for up to this many splits (one workgroup adds all slabs of its tile; long splits keep gm3d_sum_few_rows)
_nt_counters = {}          # device -> [int32 zeros, next free offset]: tile counters, self-resetting; slices handed out round-robin


def _nt_counter_slice(device, need):
    """`need` zeroed ints no other launch in flight uses (a rotating slice of one persistent buffer; 64 Ki counters = > 100 launches
    of the step's largest product before a slice comes round again).  None while a stream capture runs and the buffer does not exist
    yet (a buffer first zeroed inside a capture holds nothing until the first replay)."""
    key = str(device)
    ent = _nt_counters.get(key)
    if ent is None:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return None
        ent = _nt_counters[key] = [torch.zeros(1 << 16, dtype=torch.int32, device=device), 0]
    buf, off = ent
    if need > buf.numel():
        return None
    if off + need > buf.numel():
        off = 0
    ent[1] = off + need
    return buf[off:off + need]


from . import streams as _streams_mod  # noqa: E402
_streams_mod.before_capture(lambda device: _nt_counter_slice(device, 0))   # the counters exist before any audited capture begins


def wgrad_nt(dy, x, out=None, splits=None):
    """out (nb,N,K) f32 = dy[b]^T @ x[b]  (dy (nb,R,N), x (nb,R,K) bf16) -- every weight gradient of a block stack in one launch,
    written into `out` (the parameters' slots of the flat gradient buffer) when given.  Long reductions over few output tiles
    are cut into `splits` row ranges whose fp32 partial products are added in slab order: inside the same launch for up to
    FUSE_SLAB_SUM_MAX splits (gm3d_gemm_nt_bf16_sum), by gm3d_sum_few_rows beyond -- the same bits either way."""
    nb, R, N = dy.shape
    K = x.shape[2]
    if splits is None:
        splits = lib.gm3d_gemm_nt_splits(nb, R, N, K)
        if WGRAD_SPLITS_CAP and nb == WGRAD_SPLITS_CAP_NB and R <= 8192:
            splits = min(splits, WGRAD_SPLITS_CAP)
    if out is None:
        out = torch.empty(nb, N, K, dtype=torch.float32, device=dy.device)
    assert out.dtype == torch.float32 and out.stride(2) == 1 and out.stride(1) >= K
    if splits == 1:
        _launch("gm3d_gemm_nt_bf16", {"B": nb, "M": R, "N": N, "K": K}, lib.gm3d_gemm_nt_bf16, _ptr(dy), _ptr(x), _ptr(out), nb, R, N, K,
                dy.stride(1), x.stride(1), out.stride(1), dy.stride(0), x.stride(0), out.stride(0), 1, 0, _stream())
        return out
    part = torch.empty(nb, splits, N, K, dtype=torch.float32, device=dy.device)
    if (FUSE_SLAB_SUM and splits <= FUSE_SLAB_SUM_MAX and K % 4 == 0 and out.stride(1) % 4 == 0 and out.data_ptr() % 16 == 0
            and (nb == 1 or out.stride(0) >= N * out.stride(1))):
        cnt = _nt_counter_slice(dy.device, nb * lib.gm3d_gemm_nt_tiles(N, K))
        if cnt is not None:
            _launch("gm3d_gemm_nt_bf16_sum", {"B": nb, "M": R, "N": N, "K": K}, lib.gm3d_gemm_nt_bf16_sum, _ptr(dy), _ptr(x), _ptr(part),
                    _ptr(out), _ptr(cnt), nb, R, N, K, dy.stride(1), x.stride(1), out.stride(1), dy.stride(0), x.stride(0), out.stride(0),
                    splits, _stream())
            return out
    _launch("gm3d_gemm_nt_bf16", {"B": nb, "M": R, "N": N, "K": K}, lib.gm3d_gemm_nt_bf16, _ptr(dy), _ptr(x), _ptr(part), nb, R, N, K,
            dy.stride(1), x.stride(1), K, dy.stride(0), x.stride(0), splits * N * K, splits, N * K, _stream())
    if out.is_contiguous() or (out.stride(1) == K and out.stride(0) == N * K):
        _launch("gm3d_sum_few_rows", {"rows": nb * splits, "cols": N * K}, lib.gm3d_sum_few_rows, _ptr(part), nb, splits, N * K, _ptr(out),
                _stream())
    else:
        out.copy_(part.sum(dim=1))
    return out


MULTI_TARGET_WGS = 512    # ... and for a launch with few tiles: the workgroup count its row splits aim at
MULTI_SPLITS_MAX = 2      # with MULTI_WGRAD: cap on the row splits of the 8192-row problems (1 for <= 4096 rows) when the whole launch has >= 768 tiles:
#                           the launch as a whole fills the chip, so a problem needs splits only against a long tail (0 = keep each problem's own)
MULTI_WGRAD = True        # the weight gradients of several stacks as ONE launch (+ one slab-sum launch): gm3d_gemm_nt_bf16_multi.  With each
#                           problem's own splits: 22 launches fewer and no faster (7.64 vs 7.60 ms same-box); with the splits capped as above (fewer
#                           slabs to write and add): 7.56 vs 7.65 ms, +1.1 %


NT_TN_FASTEST = False     # measurement knob: tile order inside a (block, row-split) group of the one-launch weight gradients (csrc/gemm_nt.hip
#                           gm3d_gemm_nt_set_order): PMC and step time equal either way (profiles/NEGATIVE_RESULTS.md, round 4)
_nt_order_set = [False]


def wgrad_multi_ok(dy, x, out):
    """a request wgrad_nt_multi takes: what wgrad_nt takes, with a destination of dense (N,K) matrices at a constant batch stride"""
    nb, R, N = dy.shape
    K = x.shape[2]
    return (wgrad_supported(dy, x) and K % 4 == 0
            and (out is None or (out.dtype == torch.float32 and out.shape == (nb, N, K) and out.stride(2) == 1 and out.stride(1) == K
                                 and (nb == 1 or out.stride(0) >= N * K) and out.data_ptr() % 16 == 0)))


def wgrad_nt_multi(reqs, splits=None, want_splits=False):
    """[(dy (nb,R,N), x (nb,R,K), out (nb,N,K) f32 | None), ...] (at most 16) -> [out, ...]: every product dy[b]^T @ x[b] of every request
    in ONE launch of the 128 x 128-tile weight-gradient kernel, then ONE launch that adds the row-split slabs (csrc/gemm_nt.hip
    gm3d_gemm_nt_bf16_multi).  Bit-identical to wgrad_nt per request with the same row splits (splits=: one per request; default: each
    request's own choice, capped by MULTI_SPLITS_MAX when the launch is large).
    NOTE: the DEFAULT splits of a request depend on the composition of the whole launch (its total tile count), and the
    fp32 summation order of a weight gradient follows its splits.  The same product batched differently -- a stack alone, with the
    deferred decoders, inside SegmentedDDPStep's cut, per-request fallback beyond 16 -- may therefore differ in the last bits
    (tests/test_gpu_gemm.py::test_wgrad_multi_default_splits_depend_on_the_batch_only_in_the_last_bits pins the size of that: 1e-6)."""
    import ctypes
    n = len(reqs)
    VP, I, LL = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_longlong * n
    outs, parts, spl = [], [], []
    tiles = [r[0].shape[0] * ((r[0].shape[2] + 127) // 128) * ((r[1].shape[2] + 127) // 128) for r in reqs]
    tiles_all = sum(tiles)
    rows_target = max(512, sum(t_ * r[0].shape[1] for t_, r in zip(tiles, reqs)) // MULTI_TARGET_WGS)
    for dy, x, out in reqs:
        nb, R, N = dy.shape
        K = x.shape[2]
        if out is None:
            out = torch.empty(nb, N, K, dtype=torch.float32, device=dy.device)
        s_ = lib.gm3d_gemm_nt_splits(nb, R, N, K)
        if splits is not None:
            s_ = splits[len(spl)]
        elif MULTI_SPLITS_MAX and tiles_all < 768:
            # few tiles in all (the mini-PointNet's products: 2-12 tiles each over 8192 .. 262,144 rows): splits per problem in
            # proportion to its rows, so that the launch has ~512 workgroups of about equal length (each problem on its own would cut
            # itself into 64 slabs to fill the chip alone)
            s_ = 1
            while 2 * s_ <= min(64, R // rows_target) and R % (64 * s_) == 0:
                s_ *= 2
        elif MULTI_SPLITS_MAX and tiles_all >= 768:
            # the launch as a whole fills the chip many times over: a problem needs row splits only to keep its tiles from running
            # much longer than the others' (tail), not to fill CUs -- fewer slabs to write and add
            s_ = min(s_, MULTI_SPLITS_MAX if R > 4096 else 1)
            while s_ > 1 and R % (32 * s_):
                s_ -= 1
        outs.append(out)
        spl.append(s_)
        parts.append(torch.empty(nb, s_, N, K, dtype=torch.float32, device=dy.device) if s_ > 1 else None)
    if _nt_order_set[0] != bool(NT_TN_FASTEST):
        lib.gm3d_gemm_nt_set_order(int(bool(NT_TN_FASTEST)))
        _nt_order_set[0] = bool(NT_TN_FASTEST)
    _launch("gm3d_gemm_nt_bf16_multi", {"count": n, "problems": [(r[0].shape[0], r[0].shape[1], r[0].shape[2], r[1].shape[2]) for r in reqs],
                                        "splits": list(spl)}, lib.gm3d_gemm_nt_bf16_multi, n,
            VP(*[_ptr(r[0]) for r in reqs]), VP(*[_ptr(r[1]) for r in reqs]), VP(*[_ptr(o) for o in outs]),
            VP(*[(_ptr(p) if p is not None else None) for p in parts]), I(*[r[0].shape[0] for r in reqs]), I(*[r[0].shape[1] for r in reqs]),
            I(*[r[0].shape[2] for r in reqs]), I(*[r[1].shape[2] for r in reqs]), I(*[r[0].stride(1) for r in reqs]),
            I(*[r[1].stride(1) for r in reqs]), LL(*[r[0].stride(0) for r in reqs]), LL(*[r[1].stride(0) for r in reqs]),
            LL(*[o.stride(0) for o in outs]), I(*spl), _stream())
    return (outs, spl) if want_splits else outs


def tile_rows(M):
    return lib.gm3d_gemm_tile_rows(int(M))


def linear_gelu_bwd(d_o, w2t, f, bias, df, colpart):
    """df = (d_o @ w2t^T) * GELU'(f + bias); w2t (hidden, C) = fc2.weight^T in bf16; colpart (tile_rows(M), hidden) f32."""
    M, K = d_o.shape
    N = w2t.shape[0]
    _launch("gm3d_gemm_tn_bf16_gelu_bwd", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_gelu_bwd, _ptr(d_o), _ptr(w2t), _ptr(f),
            _ptr(bias), _ptr(df), _ptr(colpart), M, N, K, d_o.stride(0), w2t.stride(0), f.stride(0), df.stride(0), _stream())
    return df


def linear_gelu_bwd_dma(d_o, w2t, f, bias, df, colpart, bm=64):
    """linear_gelu_bwd on csrc/gemm_dma.hip; colpart (ceil(M / bm), hidden) f32."""
    M, K = d_o.shape
    N = w2t.shape[0]
    assert colpart.shape[0] * bm >= M and colpart.shape[1] == N
    _launch("gm3d_gemm_tn_bf16_dma_gelu_bwd", {"M": M, "N": N, "K": K}, lib.gm3d_gemm_tn_bf16_dma_gelu_bwd, _ptr(d_o), _ptr(w2t),
            _ptr(f), _ptr(bias), _ptr(df), _ptr(colpart), M, N, K, d_o.stride(0), w2t.stride(0), f.stride(0), df.stride(0), int(bm),
            _stream())
    return df


MULTI_TRANSPOSE = True    # the four transposed weight shadows of a stack's backward in ONE launch (gm3d_transpose_bf16_multi)


def stacked_transposes(groups):
    """[ws_0, ws_1, ...] (each a list of same-shaped (N,K) bf16 weights at a constant stride in one buffer) -> [stacked_transpose(ws_i)]:
    ONE launch for all groups when every group has the regular layout, else one per group."""
    import ctypes
    ok = MULTI_TRANSPOSE and 1 < len(groups) <= 8
    metas = []
    for ws in groups:
        w0 = ws[0]
        n = len(ws)
        step = (ws[1].data_ptr() - w0.data_ptr()) if n > 1 else w0.numel() * w0.element_size()
        regular = (w0.dtype == torch.bfloat16 and w0.is_cuda and w0.dim() == 2 and w0.shape[0] % 8 == 0 and w0.shape[1] % 8 == 0
                   and w0.data_ptr() % 16 == 0 and step % 16 == 0 and step > 0 and step % w0.element_size() == 0 and step // w0.element_size() >= w0.numel()
                   and all(w.shape == w0.shape and w.is_contiguous() and w.data_ptr() - w0.data_ptr() == i * step for i, w in enumerate(ws)))
        ok = ok and regular
        metas.append((w0, n, step // w0.element_size()))
    if not ok:
        return [stacked_transpose(ws) for ws in groups]
    outs = [torch.empty(n, w0.shape[1], w0.shape[0], dtype=w0.dtype, device=w0.device) for w0, n, _ in metas]
    c = len(groups)
    VP, I, LL = ctypes.c_void_p * c, ctypes.c_int * c, ctypes.c_longlong * c
    _launch("gm3d_transpose_bf16_multi", {"count": c}, lib.gm3d_transpose_bf16_multi, c, VP(*[_ptr(m_[0]) for m_ in metas]),
            VP(*[_ptr(o) for o in outs]), I(*[m_[1] for m_ in metas]), I(*[m_[0].shape[0] for m_ in metas]),
            I(*[m_[0].shape[1] for m_ in metas]), LL(*[m_[2] for m_ in metas]), _stream())
    return outs


def stacked_transpose(ws):
    """[w_0 .. w_{n-1}] (each (N,K) bf16) -> (n, K, N) contiguous = the transposes, in ONE copy launch when the tensors sit at a
    constant stride in one buffer (the flat optimizer's bf16 shadow), two otherwise."""
    w0 = ws[0]
    n = len(ws)
    if n > 1:
        step = ws[1].data_ptr() - w0.data_ptr()
        regular = step > 0 and step % w0.element_size() == 0 and all(
            w.shape == w0.shape and w.is_contiguous() and w.untyped_storage().data_ptr() == w0.untyped_storage().data_ptr()
            and w.data_ptr() - w0.data_ptr() == i * step for i, w in enumerate(ws))
        if (regular and w0.shape[0] % 8 == 0 and w0.shape[1] % 8 == 0 and w0.dtype == torch.bfloat16 and w0.data_ptr() % 16 == 0
                and step % 16 == 0):
            N, K = w0.shape
            out = torch.empty(n, K, N, dtype=w0.dtype, device=w0.device)
            _launch("gm3d_transpose_bf16_batched", {"batch": n, "rows": N, "cols": K}, lib.gm3d_transpose_bf16_batched, _ptr(w0),
                    _ptr(out), n, N, K, step // w0.element_size(), _stream())
            return out
        if regular:
            N, K = w0.shape
            v = torch.as_strided(w0, (n, N, K), (step // w0.element_size(), K, 1))
            return v.transpose(1, 2).contiguous()
    return torch.stack(list(ws)).transpose(1, 2).contiguous()
