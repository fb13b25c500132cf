"""qkv projection + attention: one launch (gm3d_attention_qkv_fwd) against the two it replaces, at the teacher's / student's shapes.
Captured trains of launches replayed, HIP events.    python tools/attn_qkv_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import fused, gemm
from tools.attn_bench import bench

dev = torch.device("cuda")
H, C = 6, 384
for (B, T) in ((64, 64), (128, 64), (128, 25), (128, 26)):
    def mk():
        h = torch.randn(B * T, C, device=dev).bfloat16()
        w = (torch.randn(3 * C, C, device=dev) * 0.06).bfloat16()
        return (h, w, torch.empty(B * T, 3 * C, device=dev, dtype=torch.bfloat16), torch.empty(B * T, C, device=dev, dtype=torch.bfloat16))
    sets = [mk() for _ in range(4)]
    def two(h, w, qkv, o):
        gemm.mm(h, w, out=qkv)
        fused._attention_fwd(qkv, B, T, H, 0.125, out=o)
    t2 = bench("two", two, sets)
    tg = bench("gemm", lambda h, w, qkv, o: gemm.mm(h, w, out=qkv), sets)
    t1 = bench("one", lambda h, w, qkv, o: fused._attention_qkv_fwd(h, w, B, T, H, 0.125, out=o), sets)
    t1q = bench("oneq", lambda h, w, qkv, o: fused._attention_qkv_fwd(h, w, B, T, H, 0.125, out=o, want_qkv=True, want_lse=True), sets)
    fl = 2.0 * B * T * 3 * C * C + B * H * 4.0 * T * T * 64
    print("B=%3d T=%2d  gemm(%s) %6.2f us + attention = %6.2f us   fused %6.2f us (%5.1f TFLOP/s)   fused+qkv out %6.2f us" %
          (B, T, gemm.choose(B * T, 3 * C, C), tg, t2, t1, fl / t1 * 1e-6, t1q))
