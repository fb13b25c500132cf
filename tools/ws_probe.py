"""Where does a tile of the weight-stationary GEMM (csrc/gemm_ws.hip) spend its time?  Builds the kernel with one phase compiled
out at a time (-DGM3D_WS_PROBE_NO_MFMA / _NO_STORE / _NO_LOAD: results are garbage, only the time matters) into scratch libraries
and times the 262144 x 256 -> 512 and 262144 x 512 -> 256 products.   python tools/ws_probe.py   (GPU box; ~2 min of hipcc)"""
import ctypes, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gm3d_amd", "csrc", "gemm_ws.hip")
tmp = tempfile.mkdtemp(prefix="ws_probe_")
variants = {"full": [], "no_mfma": ["-DGM3D_WS_PROBE_NO_MFMA"], "no_store": ["-DGM3D_WS_PROBE_NO_STORE"], "no_load": ["-DGM3D_WS_PROBE_NO_LOAD"],
            "no_mfma_no_store": ["-DGM3D_WS_PROBE_NO_MFMA", "-DGM3D_WS_PROBE_NO_STORE"],
            "nothing": ["-DGM3D_WS_PROBE_NO_MFMA", "-DGM3D_WS_PROBE_NO_STORE", "-DGM3D_WS_PROBE_NO_LOAD"]}
libs = {}
for name, flags in variants.items():
    so = os.path.join(tmp, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", src, "-o", so] + flags)
    libs[name] = ctypes.CDLL(so)
vp, i32 = ctypes.c_void_p, ctypes.c_int
for M, K, N in ((262144, 256, 512), (262144, 512, 256), (262144, 128, 256)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = "M=%d K=%d N=%d:" % (M, K, N)
    for name, lib in libs.items():
        f = lib.gm3d_gemm_tn_bf16_ws
        f.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            assert f(x.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, K, K, N, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f(x.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, K, K, N, st)
        e1.record()
        torch.cuda.synchronize()
        line += "  %s %.1f us" % (name, e0.elapsed_time(e1) * 100)
    print(line)
