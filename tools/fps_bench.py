import torch, sys, os
sys.path.insert(0, "/root/repo")
from gm3d_amd import ops
from bench import make_clouds
x = make_clouds(32, 8192, 1, torch.device("cuda"))
for _ in range(3): ops.fps(x, 1200)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.fps(x, 1200)
e1.record(); torch.cuda.synchronize()
t8 = e0.elapsed_time(e1) * 100
x = make_clouds(128, 1024, 1, torch.device("cuda"))
for _ in range(3): ops.fps(x, 64)
torch.cuda.synchronize()
e0.record()
for _ in range(20): ops.fps(x, 64)
e1.record(); torch.cuda.synchronize()
print("cfg", os.environ.get("GM3D_FPS_CFG"), "fps 1024->64 B=128: %.1f us" % (e0.elapsed_time(e1) * 50), end="; ")
print("cfg", os.environ.get("GM3D_FPS_CFG"), "fps 8192->1200 B=32: %.1f us" % t8)

# Point-M2AE three-level grouping (config #4: 128 clouds of 2048 points)
from gm3d_amd.hierarchical_group import HierarchicalGroup
hg = HierarchicalGroup()
x = make_clouds(128, 2048, 2, torch.device("cuda"))
for _ in range(3): hg(x)
torch.cuda.synchronize()
e0.record()
for _ in range(10): hg(x)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 100
print("M2AE 3-level grouping B=128 N=2048 (512x16, 256x8, 64x8): %.1f us -> %.0f clouds/s" % (t, 128 / t * 1e6))
