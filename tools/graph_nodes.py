"""What kinds of nodes the captured step graphs hold (hipGraphDebugDotPrint through torch's CUDAGraph.debug_dump): kernel / memcpy / memset
counts per graph of the single-graph step, the four-graph data-parallel step and the Point-M2AE step.  Memset nodes misbehaved on
replays on this stack (csrc/chamfer.hip zero_f32_kernel); this tool is how their absence is checked.
    python tools/graph_nodes.py [outdir]"""
import os, re, sys, collections
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/graph_nodes"
os.makedirs(out, exist_ok=True)
torch.cuda.set_device(0)
_Base = torch.cuda.CUDAGraph
made = []


class Dumped(_Base):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, True)          # keep_graph: the hipGraph_t survives capture_end, debug_dump can print it
        g.enable_debug_mode()
        made.append(g)
        return g

    def __init__(self, *a, **k):
        super().__init__(True)                  # (the binding reads keep_graph in __init__)
        self.enable_debug_mode()


torch.cuda.CUDAGraph = Dumped
torch.cuda.graphs.CUDAGraph = Dumped
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M
import bench
dev = torch.device("cuda", 0)
args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=True, accum_iter=1, lr=1e-3, min_lr=0.0, warmup_epochs=40)
data = bench.make_clouds(int(os.environ.get("B", "128")), 1024, 1234, dev)


KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "child graph", 5: "empty", 6: "wait event", 7: "event record", 10: "mem alloc",
         11: "mem free"}          # hipGraphNodeType
_hip = None


def census(tag):
    """node kinds of every captured graph (hipGraphGetNodes / hipGraphNodeGetType on the kept hipGraph_t)"""
    global _hip
    import ctypes
    if _hip is None:
        _hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    for k, g in enumerate(made):
        try:
            raw = ctypes.c_void_p(g.raw_cuda_graph())
        except RuntimeError as ex:
            print("%-10s graph %d  (%s)" % (tag, k, str(ex).splitlines()[0][:80]))
            continue
        n = ctypes.c_size_t(0)
        assert _hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
        nodes = (ctypes.c_void_p * max(n.value, 1))()
        assert _hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
        kinds = collections.Counter()
        for i in range(n.value):
            t = ctypes.c_int(-1)
            assert _hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) == 0
            kinds[KINDS.get(t.value, "type %d" % t.value)] += 1
        print("%-10s graph %d  %d nodes  %s" % (tag, k, n.value, dict(kinds)))
    made.clear()


def build(segmented):
    torch.manual_seed(0)
    m = M.mae_vit_base_patch16_dec512d8b(norm_pix_loss=False).to(dev).train()
    ema = E.ModelEma(m, decay=0.999)
    opt = E.build_optimizer(m, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema, segment_of=E.ddp_segment if segmented else None)
    E.adjust_learning_rate(opt, 200.0, args)
    return (E.SegmentedDDPStep if segmented else E.GraphedPretrainStep)(m, ema, opt, args, data, 200)


s = build(False); census("single"); del s
s = build(True); census("segmented"); del s
if os.environ.get("M2AE", "1") == "1":
    sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "1"]
    import runpy
    try:
        runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_m2ae.py"), run_name="__main__")
    except SystemExit:
        pass
    census("m2ae")
