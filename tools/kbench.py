"""Micro-benchmark of the hand-written streaming kernels at the shapes of the B=128 pretrain step.
Each kernel runs as a captured train of launches (hipGraph replay) on a rotating set of NSET buffers (6: larger than
the 256 MB Infinity Cache; NSET=1: cache-warm), timed with HIP events; GB/s = algorithmic bytes (bench.algorithmic) / time.   python tools/kbench.py [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd._capi import lib, check
from gm3d_amd.ops import _ptr, _stream, _DT
from bench import algorithmic

dev = torch.device("cuda")
bf = torch.bfloat16
flt = sys.argv[1] if len(sys.argv) > 1 else ""
NSET = int(os.environ.get("NSET", 6))


def run(name, meta, make, call, iters=60):
    if flt and flt not in name:
        return
    sets = [make() for _ in range(NSET)]
    for i in range(NSET):
        call(*sets[i])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()          # replay a captured train of launches: the host cannot issue one per ~5 us
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for i in range(iters):
                call(*sets[i % NSET])
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (2 * iters)
    bound, amount, unit = algorithmic(name, meta)
    rate = amount / us * 1e-3 if unit == "B" else amount / us * 1e-6
    print("%-28s %-44s %8.1f us  %8.1f %s  (%.0f%% of %s)" % (name, str({k: v for k, v in meta.items() if k != "dtype"}), us, rate,
          "GB/s" if unit == "B" else "TFLOP/s", 100 * rate / (8000 if unit == "B" else 2500), "8 TB/s" if unit == "B" else "2.5 PF"))


def r(*shape, dtype=bf):
    return torch.randn(*shape, device=dev, dtype=torch.float32).to(dtype)


for R in (3200, 8192):
    C = 384
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rsc = torch.ones(128, device=dev)
    run("gm3d_residual_ln_fwd", {"R": R, "dtype": str(bf)},
        lambda: (r(R, C, dtype=torch.float32), r(R, C), r(R, C), torch.empty(R, C, device=dev), torch.empty(R, C, device=dev, dtype=bf),
                 torch.empty(R, device=dev), torch.empty(R, device=dev)),
        lambda res, y, add, ores, h, mu, rs: check(lib.gm3d_residual_ln_fwd(_ptr(res), _ptr(y), _ptr(b), _ptr(rsc), R // 128, _ptr(add),
                                                                            _ptr(g), _ptr(b), 1e-5, _ptr(ores), _ptr(h), _ptr(mu), _ptr(rs),
                                                                            R, C, 1, _stream()), "ln_fwd"))
    nrows = lib.gm3d_ln_partial_rows(R)
    run("gm3d_residual_ln_bwd", {"R": R, "dtype": str(bf)},
        lambda: (r(R, C), r(R, C, dtype=torch.float32), r(R, C, dtype=torch.float32), torch.zeros(R, device=dev), torch.ones(R, device=dev),
                 torch.empty(R, C, device=dev), torch.empty(R, C, device=dev, dtype=bf), torch.empty(nrows, 3 * C, device=dev)),
        lambda dh, gin, x, mu, rs, dx, dy, part: check(lib.gm3d_residual_ln_bwd(_ptr(dh), _ptr(gin), _ptr(x), _ptr(mu), _ptr(rs), _ptr(g),
                                                                                 _ptr(rsc), R // 128, _ptr(dx), _ptr(dy), None, None, _ptr(part),
                                                                                 R, C, 1, _stream()), "ln_bwd"))
    C4 = 1536
    b4 = torch.zeros(C4, device=dev)
    run("gm3d_bias_gelu_fwd", {"R": R, "C": C4, "dtype": str(bf)},
        lambda: (r(R, C4), torch.empty(R, C4, device=dev, dtype=bf)),
        lambda f, o: check(lib.gm3d_bias_gelu_fwd(_ptr(f), _ptr(b4), _ptr(o), R, C4, 1, _stream()), "gelu_fwd"))
    nrg = lib.gm3d_gelu_partial_rows(R)
    run("gm3d_bias_gelu_bwd", {"R": R, "C": C4, "dtype": str(bf)},
        lambda: (r(R, C4), r(R, C4), torch.empty(R, C4, device=dev, dtype=bf), torch.empty(nrg, C4, device=dev)),
        lambda dg, f, df, part: check(lib.gm3d_bias_gelu_bwd(_ptr(dg), _ptr(f), _ptr(b4), _ptr(df), _ptr(part), R, C4, 1, _stream()), "gelu_bwd"))

G, K = 8192, 32
for C in (256, 384, 512):
    bias = torch.zeros(C, device=dev)
    if C != 512:
        run("gm3d_group_max_fwd", {"G": G, "K": K, "C": C, "dtype": str(bf)},
            lambda: (r(G * K, C), torch.empty(G, C, device=dev, dtype=bf), torch.empty(G, C, device=dev, dtype=torch.uint8)),
            lambda x, o, a: check(lib.gm3d_group_max_fwd(_ptr(x), _ptr(bias), _ptr(o), _ptr(a), G, K, C, 1, _stream()), "gmax"))
        run("gm3d_group_max_bwd", {"G": G, "K": K, "C": C, "dtype": str(bf)},
            lambda: (r(G, C), torch.randint(0, K, (G, C), device=dev, dtype=torch.uint8), torch.empty(G * K, C, device=dev, dtype=bf)),
            lambda d, a, o: check(lib.gm3d_group_max_bwd(_ptr(d), _ptr(a), _ptr(o), G, K, C, 1, _stream()), "gmaxb"))
    if C == 512:
        sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        nr = lib.gm3d_embed_partial_rows(1, G, C)
        run("gm3d_bn_bcast_stats", {"G": G, "K": K, "C": C, "dtype": str(bf)},
            lambda: (r(G * K, C), r(G, C), torch.empty(nr, 2 * C, device=dev)),
            lambda y0, t, p: check(lib.gm3d_bn_bcast_stats(_ptr(y0), _ptr(t), G, K, C, _ptr(p), 1, _stream()), "bnstats"))
        run("gm3d_bn_bcast_apply_relu", {"G": G, "K": K, "C": C, "dtype": str(bf)},
            lambda: (r(G * K, C), r(G, C), torch.empty(G * K, C, device=dev, dtype=bf)),
            lambda y0, t, o: check(lib.gm3d_bn_bcast_apply_relu(_ptr(y0), _ptr(t), _ptr(sc), _ptr(sh), _ptr(o), G, K, C, 0.0, 1, _stream()), "bnapply"))
        run("gm3d_bn_bcast_bwd_stats", {"G": G, "K": K, "C": C, "dtype": str(bf)},
            lambda: (r(G * K, C), r(G * K, C), r(G, C), torch.empty(nr, 2 * C, device=dev)),
            lambda da, y0, t, p: check(lib.gm3d_bn_bcast_bwd_stats(_ptr(da), _ptr(y0), _ptr(t), _ptr(sc), _ptr(sh), _ptr(sh), _ptr(sc), G, K, C,
                                                                   _ptr(p), 0.0, 1, _stream()), "bnbs"))
        run("gm3d_bn_bcast_bwd_apply", {"G": G, "K": K, "C": C, "dtype": str(bf)},
            lambda: (r(G * K, C), r(G * K, C), r(G, C), torch.empty(G * K, C, device=dev, dtype=bf), torch.empty(G, C, device=dev)),
            lambda da, y0, t, dy, dt: check(lib.gm3d_bn_bcast_bwd_apply(_ptr(da), _ptr(y0), _ptr(t), _ptr(sc), _ptr(sh), _ptr(sh), _ptr(sc),
                                                                        _ptr(sh), _ptr(sh), _ptr(dy), _ptr(dt), G, K, C, 0.0, 1, _stream()), "bnba"))
