"""fc1 + GELU (forward) and fc2 input gradient + GELU backward: register-prefetch kernel (csrc/gemm.hip) against the 192-column
LDS-DMA kernel (csrc/gemm_dma.hip, 64- / 128-row tiles) at the step's row counts.   python tools/gelu_kbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm
from tools.gemm_kbench import train

dev = torch.device("cuda")
if __name__ == "__main__":
    K, N = 384, 1536
    for M in (3200, 4096, 8192):
        sets = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(), torch.randn(N, device=dev),
                 torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16),
                 torch.empty((M + 63) // 64, N, device=dev)) for _ in range(4)]
        for want_f in (False, True):
            t0 = train(lambda x, w, b, f, g, cp: gemm.linear_gelu(x, w, b, f_out=f if want_f else None, g_out=g), sets)
            t1 = train(lambda x, w, b, f, g, cp: gemm.linear_gelu_dma(x, w, b, f_out=f if want_f else None, g_out=g, bm=64), sets)
            t2 = train(lambda x, w, b, f, g, cp: gemm.linear_gelu_dma(x, w, b, f_out=f if want_f else None, g_out=g, bm=128), sets)
            print("fwd M=%5d f_out=%d   own %6.1f us   dma64 %6.1f us   dma128 %6.1f us" % (M, want_f, t0, t1, t2))
        t0 = train(lambda x, w, b, f, g, cp: gemm.linear_gelu_bwd(x, w, f, b, g, cp), sets)
        t1 = train(lambda x, w, b, f, g, cp: gemm.linear_gelu_bwd_dma(x, w, f, b, g, cp, bm=64), sets)
        t2 = train(lambda x, w, b, f, g, cp: gemm.linear_gelu_bwd_dma(x, w, f, b, g, cp, bm=128), sets)
        print("bwd M=%5d           own %6.1f us   dma64 %6.1f us   dma128 %6.1f us" % (M, t0, t1, t2))
