"""Run the own GEMM on one shape a few times (for rocprofv3 counter passes): python tools/gemm_one.py M K N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm
M, K, N = (int(v) for v in sys.argv[1:4])
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(10):
    gemm.linear_tn(x, w, out=o)
torch.cuda.synchronize()
