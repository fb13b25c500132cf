"""Captured step vs eager step, parameter by parameter (gpurun_out/r2i showed a 2.3 % gap in segment 0).

B clouds, DropPath off, mask noise injected, bf16.  From ONE model state the three backward segments of SegmentedDDPStep run
  (a) eagerly, twice            -> run-to-run differences of eager execution
  (b) as captured graphs, twice -> replay-to-replay differences
and the flat gradient buffers are compared slot by slot, names printed for everything that differs.  Then the same for the
single-graph GraphedPretrainStep against step_forward_backward.  Finally every library call left in the step is run eagerly
and inside a capture on identical operands (does the library pick another solution under capture?).

    python tools/eager_vs_graph_diag.py [B]
"""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gm3d_amd import engine_pretrain as E  # noqa: E402
from gm3d_amd import models_mae_learn_loss as M  # noqa: E402
from tests import clouds  # noqa: E402


def build(segmented=True):
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    for mod in m.modules():
        if isinstance(mod, M.DropPath):
            mod.drop_prob = 0.0
    ema = E.ModelEma(m, 0.999)
    opt = E.build_optimizer(m, lr=2e-4, flat=True, model_ema=ema, segment_of=E.ddp_segment if segmented else None)
    return m, ema, opt


def report(tag, a, b, opt, top=12):
    if torch.equal(a, b):
        print("%-44s identical" % tag)
        return
    offs = list(opt._offs) + [opt.n]
    rows = []
    for (name, p), o, e in zip(opt._named, offs[:-1], offs[1:]):
        e = o + p.numel()
        d = float((a[o:e] - b[o:e]).abs().max())
        if d > 0:
            rows.append((d / max(float(b[o:e].abs().max()), 1e-30), d, name))
    rows.sort(reverse=True)
    print("%-44s %d of %d tensors differ; worst relative (to the tensor's max):" % (tag, len(rows), len(opt._named)))
    for r, d, n in rows[:top]:
        print("      %.3e  (abs %.3e)  %s" % (r, d, n))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=2e-4, min_lr=0.0, warmup_epochs=40)
    x = clouds.gaussian(B, 1024, 300).cuda()
    noise = torch.rand(B, 64, generator=torch.Generator().manual_seed(400)).cuda()

    # ---- segmented step --------------------------------------------------------------------------------------------
    m, ema, opt = build(True)
    seg = E.SegmentedDDPStep(m, ema, opt, args, x, 200, warmup_iters=0, augment=False, inject_mask_noise=True, use_graphs=False)
    seg.static_noise.copy_(noise)
    bufs = [t.detach().clone() for t in m.buffers()]

    def restore():
        with torch.no_grad():
            for t, v in zip(m.buffers(), bufs):
                t.copy_(v)

    def eager_once():
        restore()
        out = seg._phase1(x.clone())
        seg._phase2()
        seg._phase3()
        seg._cut1 = seg._cut2 = seg._cut3 = None
        torch.cuda.synchronize()
        return opt.G.clone(), {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}

    for _ in range(3):
        eager_once()            # warm-up (lazy initialisation, library workspaces)
    ge1, oe1 = eager_once()
    ge2, oe2 = eager_once()
    report("segmented eager run 1 vs run 2", ge1, ge2, opt)
    restore()
    segg = E.SegmentedDDPStep(m, ema, opt, args, x, 200, warmup_iters=0, augment=False, inject_mask_noise=True, use_graphs=True,
                              broadcast=False)
    segg.static_noise.copy_(noise)

    def graph_once():
        restore()
        segg.static_in.copy_(x)
        for k in range(3):
            segg.graphs[k].replay()
        torch.cuda.synchronize()
        return opt.G.clone(), {k: v.clone() for k, v in segg.out.items() if torch.is_tensor(v)}

    gg1, og1 = graph_once()
    gg2, og2 = graph_once()
    report("segmented graph replay 1 vs replay 2", gg1, gg2, opt)
    report("segmented graph vs eager", gg1, ge1, opt)
    for k in oe1:
        if k in og1 and oe1[k].dtype.is_floating_point:
            d = float((oe1[k].float() - og1[k].float()).abs().max())
            print("      output %-18s max |eager - graph| = %.3e%s" % (k, d, "" if d else "  (identical)"))
        elif k in og1:
            print("      output %-18s %s" % (k, "identical" if torch.equal(oe1[k], og1[k]) else "DIFFERS"))
    del seg, segg

    # ---- single-graph step -----------------------------------------------------------------------------------------
    m, ema, opt = build(False)
    bufs[:] = [t.detach().clone() for t in m.buffers()]

    def eager_fb():
        restore()
        E.step_forward_backward(m, ema, x.clone(), 200, args, optimizer=opt, augment=False, mask_noise=noise)
        opt.gather_grads()
        torch.cuda.synchronize()
        return opt.G.clone()

    for _ in range(3):
        eager_fb()
    e1, e2 = eager_fb(), eager_fb()
    report("whole-step eager run 1 vs run 2", e1, e2, opt)
    restore()
    static_x, static_n = x.clone(), noise.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        E.step_forward_backward(m, ema, static_x, 200, args, optimizer=opt, augment=False, mask_noise=static_n)
        opt.gather_grads()
    res = []
    for _ in range(2):
        restore()
        g.replay()
        torch.cuda.synchronize()
        res.append(opt.G.clone())
    report("whole-step graph replay 1 vs replay 2", res[0], res[1], opt)
    report("whole-step graph vs eager", res[0], e1, opt)

    # ---- library products, eager vs captured, on the shapes the step still hands to torch ---------------------------------
    print("library products eager vs captured (same operands):")
    R = B * 64 * 32
    shapes = [("embed y0  f @ W3l^T", (R, 256), (512, 256), True), ("embed t   fg @ W3g^T", (B * 64, 256), (512, 256), True),
              ("embed da2 dz @ W4", (B * 25 * 32, 384), (384, 512), False), ("embed df  dy @ W3l", (R, 512), (512, 256), False),
              ("embed dfg dt @ W3g", (B * 64, 512), (512, 256), False), ("embed da1 df @ W2", (R, 256), (256, 128), False),
              ("head  384->1024", (B * 64, 384), (1024, 384), True), ("head  dy @ W (1024->384)", (B * 64, 1024), (1024, 384), False),
              ("recon 384->96", (B * 64, 384), (96, 384), True), ("recon dy @ W (96->384)", (B * 64, 96), (96, 384), False)]
    gen = torch.Generator(device="cuda").manual_seed(1)
    for name, xs, ws, tn in shapes:
        a = torch.randn(*xs, device="cuda", generator=gen).bfloat16()
        w = (torch.randn(*ws, device="cuda", generator=gen) * 0.05).bfloat16()
        f = (lambda: a @ w.t()) if tn else (lambda: a @ w)
        y_e = f()
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            y_g = f()
        gg.replay()
        torch.cuda.synchronize()
        d = float((y_e.float() - y_g.float()).abs().max())
        print("      %-28s %-16s %s" % (name, tuple(xs), "identical" if d == 0 else "DIFFERS max %.3e" % d))


if __name__ == "__main__":
    main()
