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
          # mini-PointNet over the point rows (forward, input gradients; 102,400 = the student's visible rows)
          (262144, 128, 256), (262144, 256, 512), (262144, 512, 384), (102400, 512, 384), (102400, 384, 512), (262144, 512, 256), (262144, 256, 128),
          # per-group / per-token products of the embed and the heads
          (8192, 256, 512), (8192, 512, 256), (8192, 128, 384), (8192, 384, 128), (8192, 384, 1024), (8192, 1024, 384), (8192, 128, 384)]
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
    import sys
    only_new = "--new" in sys.argv
    if "--n384" in sys.argv:
        SHAPES[:] = [(M, K, 384) for M in (3200, 4096, 8192) for K in (384, 1152, 1536)]
    for M, K, N in SHAPES:
        if only_new and M in (4096, 3200) or (only_new and M == 8192 and (K, N) in ((384, 1152), (384, 384), (384, 1536), (1536, 384), (1152, 384))):
            continue
        nset = NSET if M < 100000 else 2
        sets = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(),
                 torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]
        fl = 2.0 * M * K * N
        res = {"lib": train(lambda x, w, o: torch.mm(x, w.t(), out=o), sets)}
        if K // 64 in gemm.OWN_KT:
            res["own"] = train(lambda x, w, o: gemm.linear_tn(x, w, out=o), sets)
        for bm in (64, 128):
            for depth in ((2, 3, 4) if bm == 64 else (2, 3)):
                gemm.lib.gm3d_gemm_ring_set_depth(bm, depth)
                res["ring%d%s" % (bm, "" if depth == 2 else "d%d" % depth)] = train(lambda x, w, o: gemm.linear_tn_ring(x, w, out=o, bm=bm), sets)
            gemm.lib.gm3d_gemm_ring_set_depth(bm, 2)
            if N % 96 == 0:
                res["r96_%d" % bm] = train(lambda x, w, o: gemm.linear_tn_ring96(x, w, out=o, bm=bm), sets)
            for bn in (128, 192, 256):
                if N % bn == 0:
                    res["dma%dx%d" % (bm, bn)] = train(lambda x, w, o: gemm.linear_tn_dmaw(x, w, out=o, bm=bm, bn=bn), sets)
        if M >= 32768 and gemm.lib.gm3d_gemm_ws_supported(N, K, 0):
            for occ in (1, 2):
                gemm.lib.gm3d_gemm_ws_set_occupancy(occ)
                res["ws%d" % occ] = train(lambda x, w, o: gemm.linear_tn_ws(x, w, out=o), sets)
        best = min((v, k) for k, v in res.items() if k != "lib")
        hbm = 2.0 * (M * K + M * N + N * K) / 1e6          # MB: A once, C once, W once
        print("M=%6d K=%4d N=%4d  lib %7.1f us | " % (M, K, N, res["lib"]) + "  ".join("%s %.1f" % (k, v) for k, v in res.items() if k != "lib")
              + " | best %s %.1f us = %.0f TF/s, %.2f TB/s of A+C+W, x%.2f vs lib; table says %s"
              % (best[1], best[0], fl / best[0] / 1e6, hbm / best[0], res["lib"] / best[0], gemm.choose(M, N, K)))
