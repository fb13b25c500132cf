"""One shape on the weight-stationary GEMM, a few launches (for rocprofv3 --pmc passes): python tools/ws_one.py M K N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import gemm
M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (262144, 256, 512)
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    gemm.linear_tn_ws(x, w, out=o)
torch.cuda.synchronize()
