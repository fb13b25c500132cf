"""Weight-gradient (NT) GEMM: hand-written kernel (csrc/gemm_nt.hip) vs the library path (torch.bmm -> hipBLASLt, TunableOp table)
at the step's shapes.  python tools/wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm, fused, engine_pretrain as E

E.enable_tuned_gemms()
dev = torch.device("cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (2 * iters)


shapes = [(12, 3200, 384, 1536), (12, 3200, 1536, 384), (12, 3200, 384, 384), (12, 3200, 1152, 384),
          (4, 8192, 384, 1536), (4, 8192, 1536, 384), (4, 8192, 384, 384), (4, 8192, 1152, 384),
          (1, 262144, 256, 128), (1, 262144, 512, 256), (1, 102400, 384, 512), (1, 8192, 512, 256), (1, 8192, 1024, 384)]
if __name__ == "__main__":
  print("%-28s %10s %10s %8s   splits" % ("(nb, R, N, K)", "own us", "library us", "ratio"))
  lib_ = __import__("gm3d_amd._capi", fromlist=["lib"]).lib
  for nb, R, N, K in shapes:
      dy = torch.randn(nb, R, N, device=dev).bfloat16()
      x = torch.randn(nb, R, K, device=dev).bfloat16()
      out = torch.empty(nb, N, K, device=dev)
      if N % 128 == 0 and K % 384 == 0:
          lib_.gm3d_gemm_nt_set_big_tiles(1)
          big = timeit(lambda: gemm.wgrad_nt(dy, x, out))
          s_big = lib_.gm3d_gemm_nt_splits(nb, R, N, K)
          lib_.gm3d_gemm_nt_set_big_tiles(0)
          print("   128x384 tiles %.1f us (%d splits, %.0f TFLOP/s incl. the slab sum)" % (big, s_big, 2.0 * nb * R * N * K / big * 1e-6))
      own = timeit(lambda: gemm.wgrad_nt(dy, x, out))
      was = gemm.OWN_WGRAD
      gemm.OWN_WGRAD = False
      lib_t = timeit(lambda: fused._wgrad_batched(dy, x, out))
      gemm.OWN_WGRAD = was
      fl = 2.0 * nb * R * N * K
      print("%-28s %10.1f %10.1f %8.2f   %d   (own %.0f TFLOP/s)" % (str((nb, R, N, K)), own, lib_t, lib_t / own,
            __import__("gm3d_amd._capi", fromlist=["lib"]).lib.gm3d_gemm_nt_splits(nb, R, N, K), fl / own * 1e-6))
