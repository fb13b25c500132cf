"""Own MFMA GEMM vs hipBLASLt (torch.mm, TunableOp table loaded) on the step's shapes; captured trains of launches, HIP events.
python tools/gemm_kbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm, engine_pretrain as E

E.enable_tuned_gemms()
dev = torch.device("cuda")
SHAPES = [(4096, 384, 1152), (4096, 384, 384), (4096, 384, 1536), (4096, 1536, 384), (4096, 1152, 384), (8192, 384, 1152), (8192, 384, 384), (8192, 384, 1536), (8192, 1536, 384), (8192, 1152, 384),
          (3200, 384, 1152), (3200, 384, 384), (3200, 384, 1536), (3200, 1536, 384), (3200, 1152, 384),
          (262144, 128, 256), (262144, 256, 512), (262144, 512, 384), (262144, 384, 512), (262144, 512, 256), (262144, 256, 128),
          (8192, 384, 1024), (8192, 1024, 384)]
NSET = 4


def train(fn, sets, iters=40):
    for s in sets:
        fn(*s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(iters):
                fn(*sets[i % len(sets)])
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * iters)


if __name__ == "__main__":
    for M, K, N in SHAPES:
        nset = NSET if M < 100000 else 2
        sets = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(),
                 torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]
        t_lib = train(lambda x, w, o: torch.mm(x, w.t(), out=o), sets)
        t_own = train(lambda x, w, o: gemm.linear_tn(x, w, out=o), sets)
        t_r64 = train(lambda x, w, o: gemm.linear_tn_ring(x, w, out=o, bm=64), sets)
        t_r128 = train(lambda x, w, o: gemm.linear_tn_ring(x, w, out=o, bm=128), sets)
        fl = 2.0 * M * K * N
        dma = ""
        if N % 192 == 0:
            t_d64 = train(lambda x, w, o: gemm.linear_tn_dma(x, w, out=o, bm=64), sets)
            t_d128 = train(lambda x, w, o: gemm.linear_tn_dma(x, w, out=o, bm=128), sets)
            dma = "   dma64 %7.1f us x%.2f   dma128 %7.1f us x%.2f" % (t_d64, t_lib / t_d64, t_d128, t_lib / t_d128)
        print("M=%6d K=%4d N=%4d   hipBLASLt %7.1f us %5.0f TF/s   own %7.1f us %5.0f TF/s x%.2f   ring64 %7.1f us x%.2f   ring128 %7.1f us x%.2f" %
              (M, K, N, t_lib, fl / t_lib / 1e6, t_own, fl / t_own / 1e6, t_lib / t_own, t_r64, t_lib / t_r64, t_r128, t_lib / t_r128) + dma)
