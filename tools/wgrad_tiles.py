"""csrc/gemm_nt.hip, kernel only (no slab sum): 128 x 128 tiles vs 128 x 384 tiles per row split.   python tools/wgrad_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd._capi import lib, check
from gm3d_amd.ops import _ptr, _stream
from tools.wgrad_bench import timeit

dev = torch.device("cuda")
for nb, R, N, K in [(4, 8192, 1536, 384), (4, 8192, 384, 1536), (4, 8192, 1152, 384), (4, 8192, 384, 384), (12, 3328, 1536, 384), (12, 3328, 384, 1536),
                    (12, 3328, 1152, 384), (12, 3328, 384, 384)]:
    dy = torch.randn(nb, R, N, device=dev).bfloat16()
    x = torch.randn(nb, R, K, device=dev).bfloat16()
    line = "%-24s" % str((nb, R, N, K))
    for big in (0, 1):
        lib.gm3d_gemm_nt_set_big_tiles(big)
        for splits in (1, 2, 4, 8, 16):
            if R % (32 * splits):
                continue
            part = torch.empty(nb, splits, N, K, device=dev)
            fn = lambda: check(lib.gm3d_gemm_nt_bf16(_ptr(dy), _ptr(x), _ptr(part), nb, R, N, K, N, K, K, R * N, R * K, splits * N * K, splits,
                                                     N * K, _stream()), "nt")
            t = timeit(fn)
            tiles = nb * splits * ((N // 128) * (K // 384) if big else (N // 128) * (K // 128))
            line += "  %s s%-2d %5.1f us %4.0f TF (%4d wg)" % ("BIG" if big else "sml", splits, t, 2.0 * nb * R * N * K / t * 1e-6, tiles)
        line += "\n" + " " * 24
    print(line)
lib.gm3d_gemm_nt_set_big_tiles(0)
