"""Do the captured backward graphs of SegmentedDDPStep give the SAME gradient on every replay while another process keeps the GPU busy?
A missing dependency between parallel branches of a graph survives quiet runs (the timing hides it) and shows as a rare mismatch when the
card is shared -- as the two-rank tests share it.  One process replays graphs 1-3 (no optimizer step: the parameters stay put) N times at
B clouds and compares the flat gradient buffer with the eagerly computed one after every replay; a child process runs eager steps of the
same model beside it for the whole time.
    python tools/replay_stress.py [B] [replays]          SET="mod.ATTR=val ..." as in tools/seg_b128_diag.py;  LOAD=0: no second process"""
import os, sys, subprocess, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
LOADER = len(sys.argv) > 1 and sys.argv[1] == "--load"


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))       # (several loaders / stress runs may share a box)
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
import importlib
if os.environ.get("DEBUG_OUT"):
    torch.cuda.memory._record_memory_history(max_entries=400000)
for item in os.environ.get("SET", "").split():
    name, val = item.split("=")
    mod, attr = name.rsplit(".", 1)
    setattr(importlib.import_module("gm3d_amd." + mod), attr, eval(val))
    print("set", name, val, flush=True)
from tests import clouds
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)


def build(B, graphs):
    data = clouds.gaussian(B, 1024, 900).cuda()
    noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(950)).cuda()
    torch.manual_seed(100)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    for mod in m.modules():
        if isinstance(mod, M.DropPath):
            mod.drop_prob = 0.0
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment)
    E.adjust_learning_rate(opt, 200.0, args)
    seg = E.SegmentedDDPStep(m, ema, opt, args, data, 200, warmup_iters=int(os.environ.get("WARM", "0")), augment=False, inject_mask_noise=True,
                             use_graphs=graphs)
    seg.static_noise.copy_(noise)
    return seg, opt, data


def eager(seg, data):
    seg._phase1(data)
    seg._phase2()
    seg._phase3()
    seg._cut1 = seg._cut2 = seg._cut3 = None


if LOADER:
    seg, opt, data = build(int(sys.argv[2]), False)
    eager(seg, data)
    torch.cuda.synchronize()
    if os.environ.get("READY_FILE"):          # tests wait for this before they start comparing
        open(os.environ["READY_FILE"], "w").write("ready")
    t_end = time.time() + float(sys.argv[3])
    n = 0
    while time.time() < t_end and not (os.environ.get("STOP_FILE") and os.path.exists(os.environ["STOP_FILE"])):
        eager(seg, data)
        torch.cuda.synchronize()
        n += 1
    print("loader: %d eager passes" % n, flush=True)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--m2ae":
    # the Point-M2AE step as ONE graph with a zero learning rate (parameters and teacher stay put): every replay on the same input and
    # mask noise must leave the same flat gradient buffer.  LOAD=1 a loader process beside it, LOAD=thread a second stream of this one.
    from gm3d_amd import point_m2ae as P
    B, N = 128, int(sys.argv[2]) if len(sys.argv) > 2 else 300
    pts = clouds.gaussian(B, 2048, seed=900).cuda()
    noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(40)).cuda()
    torch.manual_seed(11)
    m = P.PointM2AE().cuda().train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    ema = E.ModelEma(m, 1.0)                        # decay 1: the teacher stays exactly where it is
    opt = E.build_optimizer(m, lr=0.0, weight_decay=0.0, flat=True, model_ema=ema)
    a2 = SimpleNamespace(bf16=True, epochs=300)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            P.pretrain_step(m, ema, opt, pts.clone(), 100, a2, mask_noise=noise, augment=False)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = P.pretrain_step(m, ema, opt, pts, 100, a2, mask_noise=noise, augment=False)
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    ref, P0, E0 = opt.G.detach().clone(), opt.P.detach().clone(), opt.E.detach().clone()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print("quiet replay %d equal to the first: %s (teacher unchanged: %s)" % (i + 1, torch.equal(opt.G, ref), torch.equal(opt.E, E0)), flush=True)
    child, stop = None, []
    if os.environ.get("LOAD", "1") == "1":
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load", "128", os.environ.get("LOAD_S", "45")], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL)
        time.sleep(20)
    elif os.environ.get("LOAD") == "thread":
        import threading
        s2 = torch.cuda.Stream()
        big = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
        seg2, _, d2 = build(64, False)

        def busy():
            with torch.cuda.stream(s2):
                while not stop:
                    for _ in range(4):
                        (big @ big).relu_()
                        eager(seg2, d2)
                    s2.synchronize()
        th = threading.Thread(target=busy)
        th.start()
        time.sleep(2)
    offs = list(opt._offs) + [opt.G.numel()]
    bad, t0 = [], time.time()
    for i in range(N):
        g.replay()
        torch.cuda.synchronize()
        if not torch.equal(opt.G, ref):
            names = [(float((opt.G[o:e] - ref[o:e]).abs().max()), n) for (n, _), o, e in zip(opt._named, offs[:-1], offs[1:]) if not torch.equal(opt.G[o:e], ref[o:e])]
            bad.append((i, len(names), sorted(names, reverse=True)[:4]))
        if time.time() - t0 > 25:
            N = i + 1
            break
    stop.append(1)
    if os.environ.get("LOAD") == "thread":
        th.join()
    print("Point-M2AE B=%d: %d of %d replays differ from the first (load: %s); parameters unchanged: %s" % (B, len(bad), N, os.environ.get("LOAD", "1"),
                                                                                                      torch.equal(opt.P, P0)))
    for b in bad[:4]:
        print("  replay %d: %d parameters differ, worst %s" % b)
    if child is not None:
        print(child.communicate(timeout=120)[0].decode().strip())
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seg, opt, data = build(B, True)
eager(seg, data)
torch.cuda.synchronize()
ref = opt.G.detach().clone()
for g in seg.graphs[:3]:
    g.replay()
torch.cuda.synchronize()
OUT = {k: v.detach().clone() for k, v in seg.out.items() if torch.is_tensor(v)}       # forward results of a quiet replay
out_bad = {k: 0 for k in OUT}
if os.environ.get("DEBUG_OUT"):
    torch.cuda.synchronize()
    w = OUT["loss_chfr"]
    print("clone of loss_chfr right after it was made: %.10f at %d (source %.10f at %d)" % (float(w), w.data_ptr(), float(seg.out["loss_chfr"]), seg.out["loss_chfr"].data_ptr()))
    for gi, g in enumerate(seg.graphs[:3]):
        g.replay()
        torch.cuda.synchronize()
        print("  after replaying graph %d the clone holds %.10f" % (gi, float(w)))
    opt.G.zero_(); torch.cuda.synchronize()
    print("  after G.zero_ the clone holds %.10f" % float(w))
    A = w.data_ptr()
    snap = torch.cuda.memory._snapshot()
    for ev in snap["device_traces"][0]:
        if ev.get("addr") is not None and ev["addr"] <= A < ev["addr"] + max(ev.get("size", 0), 1) and ev["action"] in ("alloc", "free_requested", "free_completed"):
            fr = [f for f in ev.get("frames", []) if "/gm3d_amd/" in f["filename"] or "/tools/" in f["filename"]]
            print("   %-15s addr %d size %d stream %s  %s" % (ev["action"], ev["addr"], ev["size"], ev.get("stream"),
                                                        " <- ".join("%s:%d" % (os.path.basename(f["filename"]), f["line"]) for f in fr[:5])))
    for seg_ in snap["segments"]:
        if seg_["address"] <= A < seg_["address"] + seg_["total_size"]:
            print("   segment at %d size %d pool %s stream %s" % (seg_["address"], seg_["total_size"], seg_.get("segment_pool_id"), seg_.get("stream")))
child, stop, th = None, [], None
if os.environ.get("LOAD", "1") == "1":
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load", str(B), os.environ.get("LOAD_S", "45")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    time.sleep(20)                    # its import + model build
elif os.environ.get("LOAD") == "thread":          # the load from a second stream of THIS process: eager passes of a twin model + big products
    import threading
    s2 = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    seg2, _, d2 = build(64, False)

    def busy():
        with torch.cuda.stream(s2):
            while not stop:
                for _ in range(4):
                    (big @ big).relu_()
                    eager(seg2, d2)
                s2.synchronize()
    th = threading.Thread(target=busy)
    th.start()
    time.sleep(2)
offs = list(opt._offs) + [opt.G.numel()]
bad = []
t0 = time.time()
for i in range(N):
    opt.G.zero_() if i % 2 else opt.G.fill_(float("nan"))       # nothing may survive from the previous replay
    for g in seg.graphs[:3]:
        g.replay()
    torch.cuda.synchronize()
    for k, v in OUT.items():
        out_bad[k] += int(not torch.equal(seg.out[k], v))
    if not torch.equal(opt.G, ref):
        names = [(float((opt.G[o:e] - ref[o:e]).abs().max()), n) for (n, _), o, e in zip(opt._named, offs[:-1], offs[1:]) if not torch.equal(opt.G[o:e], ref[o:e])]
        bad.append((i, len(names), sorted(names, reverse=True)[:4]))
    if time.time() - t0 > float(os.environ.get("LOAD_S", "45")) - 22:
        N = i + 1
        break
stop.append(1)
if th is not None:
    th.join()
print("B=%d: %d of %d replays differ from the eager gradient%s" % (B, len(bad), N, " (load: %s)" % os.environ.get("LOAD", "1")))
print("forward results that differed from a quiet replay (count of replays):", out_bad)
if os.environ.get("DEBUG_OUT"):
    for gi, g in enumerate(seg.graphs[:4]):
        g.replay()
        torch.cuda.synchronize()
        a = seg.out["loss"].clone(); b = seg.out["loss_chfr"].clone(); c = seg.out["loss"].clone()
        torch.cuda.synchronize()
        print("after graph %d: loss %.10f  loss_chfr %.10f  loss again %.10f  item() %.10f %.10f" % (gi, float(a), float(b), float(c), seg.out["loss"].item(), seg.out["loss_chfr"].item()))
    for k in OUT:
        a, b = seg.out[k], OUT[k]
        print(k, a.dtype, tuple(a.shape), a.data_ptr(), a.flatten()[:2].tolist(), "| quiet", b.flatten()[:2].tolist())
for b in bad[:3]:
    print("  replay %d: %d parameters differ, worst %s" % b)
if child is not None:
    out, _ = child.communicate(timeout=120)
    print(out.decode().strip())
