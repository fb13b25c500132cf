"""Where does a tile of the weight-stationary GEMM (csrc/gemm_ws.hip) spend its time?  Builds patched scratch copies of the kernel
with one phase taken out at a time (no MFMA / no store / no load: results are garbage, only the time matters) into scratch libraries
and times the 262144 x 256 -> 512 and 262144 x 512 -> 256 products.   python tools/ws_probe.py   (GPU box; ~2 min of hipcc)"""
import ctypes, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gm3d_amd", "csrc", "gemm_ws.hip")
tmp = tempfile.mkdtemp(prefix="ws_probe_")
# The probe switches are NOT in the product kernel: each variant is a patched scratch copy of csrc/gemm_ws.hip (anchors below must
# match the source exactly; the tool stops if one does not).
PATCHES = {
    "no_load": [("                ws_glds16(A + (size_t)am * lda + col, base + 1024 * p);",
                 "                if (i < DEPTH) ws_glds16(A + (size_t)am * lda + col, base + 1024 * p);")],
    "no_mfma": [("""                    acc[t] = EPI == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][s], wreg[kt][s], acc[t], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[kt][s], fa[t][s], acc[t], 0, 0, 0);""",
                 "                    acc[t][0] += (float)fa[t][s][s] + (float)wreg[kt][s][0];")],
    "no_store": [("                        *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * ldc + n0 + 8 * chunk) = raw;",
                  "                        if (raw.x == 0x12345678u && raw.y == 0x9abcdef0u) *reinterpret_cast<uint4*>(C + (size_t)(m0 + row) * ldc + n0 + 8 * chunk) = raw;")],
}
variants = {"full": [], "no_mfma": ["no_mfma"], "no_store": ["no_store"], "no_load": ["no_load"], "no_mfma_no_store": ["no_mfma", "no_store"],
            "nothing": ["no_mfma", "no_store", "no_load"]}
text = open(src).read()
libs = {}
for name, which in variants.items():
    t = text
    for w in which:
        for a, b in PATCHES[w]:
            assert t.count(a) == 1, "probe anchor not found in gemm_ws.hip: " + a[:60]
            t = t.replace(a, b)
    scratch = os.path.join(tmp, name + ".hip")
    open(scratch, "w").write(t)
    so = os.path.join(tmp, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", os.path.dirname(src),
                           scratch, "-o", so])
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
