"""Kernel by kernel: same inputs, N launches while a second process keeps the GPU busy, output bits compared with a quiet launch.
    python tools/kernel_stress.py [rounds]            LOAD=0: no second process (control)"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
from gm3d_amd import fused, gemm, ops
from tests import clouds
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, T, C, H = 128, 64, 384, 6
R = B * T
bf = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(5)
rn = lambda *s, dt=bf, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(dt)
x16, pos16 = rn(R, C), rn(R, C)
res32 = rn(R, C, dt=torch.float32)
lnw, lnb = rn(C, dt=torch.float32, sc=0.1) + 1, rn(C, dt=torch.float32, sc=0.1)
wqkv, wproj, w1, w2 = rn(3 * C, C, sc=0.05), rn(C, C, sc=0.05), rn(4 * C, C, sc=0.05), rn(C, 4 * C, sc=0.05)
b1 = rn(4 * C, dt=torch.float32, sc=0.1)
bproj = rn(C, dt=torch.float32, sc=0.1)
h16, g16 = rn(R, C), rn(R, 4 * C)
data = clouds.gaussian(B, 1024, 900).cuda().contiguous()
cen = ops.fps(data, 64)[1]

cases = {
    "fps 1024->64": lambda: ops.fps(data, 64),
    "knn_group k=32": lambda: ops.knn_group(data, cen, 32, return_idx=True),
    "residual_ln_fwd first (x+pos)": lambda: fused.residual_ln_fwd(None, x16, None, None, T, pos16, lnw, lnb, 1e-6, bf, R),
    "residual_ln_fwd mid (res+y+b)": lambda: fused.residual_ln_fwd(res32, x16, bproj, None, T, None, lnw, lnb, 1e-6, bf, R),
    "attention_qkv_fwd": lambda: fused._attention_qkv_fwd(h16, wqkv, B, T, H, 0.125)[0],
    "mm proj 384->384": lambda: gemm.mm(h16, wproj),
    "mm fc2 1536->384": lambda: gemm.mm(g16, w2),
    "mm qkv 384->1152": lambda: gemm.mm(h16, wqkv),
    "linear_gelu_dma 384->1536": lambda: gemm.linear_gelu_dma(h16, w1, b1, f_out=None, g_out=None, bm=gemm.dma_bm(R))[1],
}
flat = lambda v: [t for t in (v if isinstance(v, (tuple, list)) else (v,)) if torch.is_tensor(t)]
quiet = {}
for k, fn in cases.items():
    quiet[k] = [t.clone() for t in flat(fn())]
torch.cuda.synchronize()
for k, fn in cases.items():
    assert all(torch.equal(a, b) for a, b in zip(flat(fn()), quiet[k])), "not repeatable even when quiet: " + k
child = None
stop = []
if os.environ.get("LOAD") == "thread":
    # the load from THIS process: a second stream kept busy by a host thread (same address space, no process switch on the card)
    import threading
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda", dtype=bf)

    def busy():
        with torch.cuda.stream(side):
            while not stop:
                for _ in range(20):
                    (a @ a).relu_()
                    ops.fps(data, 64)
                    fused._attention_qkv_fwd(h16, wqkv, B, T, H, 0.125)
                side.synchronize()
    th = threading.Thread(target=busy)
    th.start()
    time.sleep(2)
elif os.environ.get("LOAD", "1") == "1":
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "replay_stress.py"), "--load", "128", os.environ.get("LOAD_S", "60")],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    time.sleep(20)
bad = {k: 0 for k in cases}
detail = {}
t0, rounds = time.time(), 0
for i in range(N):
    for k, fn in cases.items():
        out = flat(fn())
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(out, quiet[k])):
            bad[k] += 1
            if k not in detail:
                j = next(j for j, (a, b) in enumerate(zip(out, quiet[k])) if not torch.equal(a, b))
                d = (out[j] != quiet[k][j])
                idx = d.nonzero()
                detail[k] = (j, int(d.sum()), tuple(out[j].shape), idx[:3].tolist(), idx[-1].tolist(),
                             float((out[j].float() - quiet[k][j].float()).abs().max()))
    rounds += 1
    if os.environ.get("LOAD") == "thread" and time.time() - t0 > 40:
        break
    if child is not None and time.time() - t0 > float(os.environ.get("LOAD_S", "60")) - 24:
        break
stop.append(1)
if os.environ.get("LOAD") == "thread":
    th.join()
print("%d rounds%s: launches whose output differed from the quiet launch" % (rounds, "" if child is None else " beside a second process"))
for k in cases:
    print("  %-34s %d   %s" % (k, bad[k], detail.get(k, "")))
if child is not None:
    print(child.communicate(timeout=120)[0].decode().strip())
