"""Does the next batch's grouping graph really run beside the training graph (point_m2ae.GraphedM2AEStep)?  HIP events on both streams.
python tools/m2ae_overlap_diag.py   (GPU box)"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, point_m2ae as P
from bench import make_clouds

torch.manual_seed(0)
model = P.PointM2AE().cuda().train()
ema = E.ModelEma(model, 0.999)
opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
args = SimpleNamespace(bf16=True, epochs=300)
pool = [make_clouds(128, 2048, 100 + i, "cuda") for i in range(4)]
g = P.GraphedM2AEStep(model, ema, opt, args, pool[0], 100)
for i in range(5):
    g(pool[i % 4], next_pts=pool[(i + 1) % 4])
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
# 1. each graph alone
for name, fn, stream in (("group graph alone", g.group_graph.replay, g.side), ("train graph alone", g.train_graph.replay, torch.cuda.current_stream())):
    with torch.cuda.stream(stream):
        a, b = ev(), ev()
        a.record()
        for _ in range(10):
            fn()
        b.record()
    torch.cuda.synchronize()
    print("%-22s %.3f ms" % (name, a.elapsed_time(b) / 10))
# 2. both, as a call issues them
t0, t1, s0, s1 = ev(), ev(), ev(), ev()
main = torch.cuda.current_stream()
t0.record(main)
with torch.cuda.stream(g.side):
    g.side.wait_event(t0)
    s0.record(g.side)
    g.group_graph.replay()
    s1.record(g.side)
g.train_graph.replay()
t1.record(main)
torch.cuda.synchronize()
print("side: start +%.3f ms, end +%.3f ms;  train graph end +%.3f ms" % (t0.elapsed_time(s0), t0.elapsed_time(s1), t0.elapsed_time(t1)))
# 3. steady state: 20 real calls with look-ahead, 20 without, 20 replays of the training graph alone
import time
def loop(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("calls with look-ahead   %.3f ms" % loop(lambda i: g(pool[i % 4], next_pts=pool[(i + 1) % 4])))
print("calls without           %.3f ms" % loop(lambda i: g(pool[i % 4])))
print("training graph only     %.3f ms" % loop(lambda i: g.train_graph.replay()))
def both(i):
    with torch.cuda.stream(g.side):
        g.group_graph.replay()
    g.train_graph.replay()
print("both graphs, no copies  %.3f ms" % loop(both))
def copies(i):
    torch._foreach_copy_(g._train_in, g._stage_out)
    g.train_graph.replay()
print("copies + training graph %.3f ms" % loop(copies))
