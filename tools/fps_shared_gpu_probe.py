"""Until round 4 fps_kernel kept an LDS copy of the cloud and gave a few clouds a different sample now and then while ANOTHER PROCESS ran
kernels on the same GPU (tools/kernel_stress.py: 23 of 1000 launches; never with the load coming from a second stream of the same process,
never on a quiet card).  Which construct was it?  Patched scratch copies of csrc/fps.hip -- the LDS copy put back, alone and with one more
construct replaced -- each launched for a fixed time beside the loader process and compared with a quiet launch of the product kernel.   python tools/fps_shared_gpu_probe.py [seconds per variant]   (GPU box; ~2 min of hipcc)"""
import ctypes, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests import clouds
src = os.path.join(ROOT, "gm3d_amd", "csrc", "fps.hip")
tmp = tempfile.mkdtemp(prefix="fps_probe_")
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
LDS_COPY = [      # the former product kernel: an LDS copy of the cloud serves the winner's coordinates
    ("    __shared__ unsigned long long slots[2][16];\n", "    __shared__ unsigned long long slots[2][16];\n    extern __shared__ float cloud[];\n"),
    ("        tmin[i] = (in && mag > 1e-3f) ? 1e10f : -1.0f;\n",
     "        tmin[i] = (in && mag > 1e-3f) ? 1e10f : -1.0f;\n        if (in) { cloud[k * 3 + 0] = x; cloud[k * 3 + 1] = y; cloud[k * 3 + 2] = z; }\n"),
    ("        ox = p[(size_t)old * 3 + 0]; oy = p[(size_t)old * 3 + 1]; oz = p[(size_t)old * 3 + 2];\n",
     "        ox = cloud[old * 3 + 0]; oy = cloud[old * 3 + 1]; oz = cloud[old * 3 + 2];\n"),
    ("dim3(B), dim3(T), 0, st, xyz, N, npoint, idx, centers);", "dim3(B), dim3(T), (size_t)N * 3 * sizeof(float), st, xyz, N, npoint, idx, centers);"),
]
SHUFFLES = [
    ("        const float m = wave_max_f32(best);\n        const int mi = wave_min_i32(best == m ? bk : 0x7fffffff);",
     "        float m = best;\n        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));\n"
     "        int mi = best == m ? bk : 0x7fffffff;\n        for (int o = 32; o; o >>= 1) mi = min(mi, __shfl_xor(mi, o));"),
    ("        unsigned long long fk = row_max_u64(slots[j & 1][lane & 15]);     // slots >= NW stay 0",
     "        unsigned long long fk = 0ull;\n        for (int q = 0; q < 16; ++q) { const unsigned long long v = slots[j & 1][q]; fk = v > fk ? v : fk; }")]
BARRIER2 = [("        unsigned long long fk = row_max_u64(slots[j & 1][lane & 15]);     // slots >= NW stay 0",
             "        unsigned long long fk = row_max_u64(slots[j & 1][lane & 15]);\n        __syncthreads();")]
ONE_WAVE = [("    if (N <= 1024) return launch_fps<256, 4>(", "    if (N <= 1024) return launch_fps<64, 16>(")]
TWO_WAVES = [("    if (N <= 1024) return launch_fps<256, 4>(", "    if (N <= 1024) return launch_fps<128, 8>(")]
PATCHES = {
    "LDS copy of the cloud (the former product kernel)": LDS_COPY,
    "LDS copy + shuffles instead of DPP": LDS_COPY + SHUFFLES,
    "LDS copy + second barrier per step": LDS_COPY + BARRIER2,
    "LDS copy + one wave per cloud (64 x 16)": LDS_COPY + ONE_WAVE,
    "LDS copy + two waves per cloud (128 x 8)": LDS_COPY + TWO_WAVES,
}
text = open(src).read()
libs = {}
for name, patches in [("product kernel", [])] + list(PATCHES.items()):
    t = text
    for a, b in patches:
        assert t.count(a) == 1, "anchor not found in fps.hip: " + a[:70]
        t = t.replace(a, b)
    scratch = os.path.join(tmp, "v%d.hip" % len(libs))
    open(scratch, "w").write(t)
    so = scratch[:-4] + ".so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-I", os.path.dirname(src),
                           "-I", os.path.join(ROOT, "include"), scratch, "-o", so])
    libs[name] = ctypes.CDLL(so)
    print("built:", name, flush=True)
vp, i32 = ctypes.c_void_p, ctypes.c_int
torch.cuda.set_device(0)
B, N, G = 128, 1024, 64
data = clouds.gaussian(B, N, 900).cuda().contiguous()
st = torch.cuda.current_stream().cuda_stream


def run(lib):
    idx = torch.empty(B, G, dtype=torch.int32, device="cuda")
    cen = torch.empty(B, G, 3, dtype=torch.float32, device="cuda")
    lib.gm3d_fps.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    assert lib.gm3d_fps(data.data_ptr(), B, N, G, idx.data_ptr(), cen.data_ptr(), st) == 0
    return idx, cen


ref = run(libs["product kernel"])
torch.cuda.synchronize()
for name, lib in libs.items():
    a = run(lib)
    torch.cuda.synchronize()
    assert torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1]), "variant differs on a quiet card: " + name
# quiet-card cost of every variant at the step's FPS shapes (north-star 1024 -> 64; Point-M2AE 2048 -> 512 -> 256 -> 64)
for (n_pts, n_out) in ((1024, 64), (2048, 512), (512, 256), (256, 64)):
    d = clouds.gaussian(B, n_pts, 901).cuda().contiguous()
    idx = torch.empty(B, n_out, dtype=torch.int32, device="cuda")
    cen = torch.empty(B, n_out, 3, dtype=torch.float32, device="cuda")
    line = "%5d -> %3d:" % (n_pts, n_out)
    for name, lib in libs.items():
        f = lib.gm3d_fps
        for _ in range(3):
            f(d.data_ptr(), B, n_pts, n_out, idx.data_ptr(), cen.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(d.data_ptr(), B, n_pts, n_out, idx.data_ptr(), cen.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        line += "  %.1f us" % (e0.elapsed_time(e1) * 50)
    print(line + "   (" + " | ".join(libs) + ")", flush=True)
if os.environ.get("TIME_ONLY") == "1":
    sys.exit(0)
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "replay_stress.py"), "--load", "128", str(20 + SEC * len(libs) + 10)],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
time.sleep(20)
for name, lib in libs.items():
    t0, n, bad, clouds_bad = time.time(), 0, 0, 0
    while time.time() - t0 < SEC:
        a = run(lib)
        torch.cuda.synchronize()
        n += 1
        if not torch.equal(a[0], ref[0]):
            bad += 1
            clouds_bad += int((a[0] != ref[0]).any(1).sum())
    print("%-56s %6d launches beside the loader: %4d differed (%d clouds)" % (name, n, bad, clouds_bad), flush=True)
print(child.communicate(timeout=200)[0].decode().strip())
