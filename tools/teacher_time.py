"""GPU time of the EMA teacher's pass alone (embed + 12 blocks + loss-prediction decoder + head + mask), captured as its own hipGraph and
replayed, for NOGRAD_SPLIT = 1 / 2 (fused.py: the inference-only stack as parallel half-batch chains).  Says whether the parallel
branches of the captured graph really overlap outside the profiler (under rocprofv3 they mostly run one after the other).
    python tools/teacher_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gm3d_amd import engine_pretrain as E, models_mae_learn_loss as M, fused
from gm3d_amd.fused import weight_cache
from bench import make_clouds

B = 128
dev = torch.device("cuda")
torch.manual_seed(0)
model = M.mae_vit_base_patch16_dec512d8b().to(dev).train()
ema = E.ModelEma(model, 0.9999)
teacher = ema.ema
x0 = make_clouds(B, 1024, 1, dev)
weight_cache.pin(teacher)
weight_cache.refresh()
L = teacher.num_group
vis = torch.zeros(B, L, dtype=torch.bool, device=dev)


def teacher_pass(parts):
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        group = teacher.group_divider(x0)
        ids = (E._arange_ids(B, L, dev), E._arange_ids(B, 0, dev))
        out = {}
        if "embed" in parts:
            out["tok"] = teacher.encoder(group[0])
        if "all" in parts:
            o = teacher(x0, mask=vis, num_visible=L, group=group, need_pix_pred=False, ids=ids)
            out["lp"] = o["loss_pred"]
        return out


for parts in (("embed",), ("all",)):
    for ns in (1, 2):
        fused.NOGRAD_SPLIT = ns
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                teacher_pass(parts)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                teacher_pass(parts)
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        print("%-6s NOGRAD_SPLIT=%d   %.3f ms per replay (FPS + KNN included)" % (parts[0], ns, a.elapsed_time(b) / 50), flush=True)
