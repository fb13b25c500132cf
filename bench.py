#!/usr/bin/env python
"""Benchmark of the hot path: one Point-MAE + GeoMask3D pretrain step (teacher fwd -> guided mask ->
student fwd/bwd -> Chamfer + ranking loss -> clip -> AdamW -> EMA) per "step", on synthetic clouds.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher's environment: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
--gpus that disagrees with WORLD_SIZE, or exceeds the visible GPUs, exits non-zero: the line never claims ranks that did not run.

Workload (BASELINE.json configs[1]/[2]): B=128 clouds per GPU, N=1024 points, G=64 groups, k=32, d=384,
depth 12 (+2x4 decoder blocks), bf16 autocast, random-init weights (seed 0), clouds ~ U[-1,1]^3 centred and
unit-sphere normalised from seed 1234+rank, resident in HBM before the timed region.  One rank per GPU; data
parallel over clouds, one bucketed RCCL gradient all-reduce per step, weak scaling.
Rank 0 prints ONE JSON line (see the keys at the bottom); `roofline` is for the hand-written HIP kernel that
takes the most time in the step, timed with HIP events on its launch stream inside the timed region;
`cpu_baseline` is the CPU oracle (oracle/, a port -- the reference's own step cannot run on a CPU) timed on
rank 0's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_FILE = "r04_pmc_fetch_write_per_launch.json"     # tools/pmc_summary.py over the two --pmc passes of this bench (profiles/README.md)
PMC_SOURCES = "r04_pmc_sources.json"                 # {"sha16": hash of gm3d_amd/csrc at the time of those passes} -> staleness is visible
PMC_FILE_M2AE = "r04_m2ae_pmc_fetch_write_per_launch.json"   # the same two passes over tools/bench_m2ae.py (the Point-M2AE step)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense

# SURVEY.md 8(d) per-unit algorithmic figures -> per launch.  GEMM-class kernels have BOTH a flop count and an operand-byte count: the
# roof that gives the larger time is the one the kernel is graded against (a fabric-bound weight-gradient launch must
# not be priced against the matrix cores).
def _gemm_both(flop, nbytes, dtype="bf16"):
    t_mfma = flop / (MFMA_PEAK_TFLOPS[dtype] * 1e12)
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
    return {"flop": flop, "bytes": nbytes, "bound": "mfma" if t_mfma >= t_hbm else "hbm"}


def algorithmic_both(name, meta):
    """-> {"flop", "bytes", "bound"} for a GEMM-class kernel (bf16 operands, fp32 weight gradients), else None."""
    if name.startswith("gm3d_gemm_tn_bf16"):   # A (M,K) + W (N,K) in, C (M,N) out, bf16
        M, N, K = meta["M"], meta["N"], meta["K"]
        return _gemm_both(2.0 * M * N * K, 2.0 * (M * K + N * K + M * N))
    if name == "gm3d_gemm_nt_bf16":            # dY (B,R,N) + X (B,R,K) in (bf16), dW (B,N,K) out (f32)
        B, R, N, K = meta["B"], meta["M"], meta["N"], meta["K"]
        return _gemm_both(2.0 * B * R * N * K, B * (2.0 * R * (N + K) + 4.0 * N * K))
    if name == "gm3d_gemm_nt_bf16_multi" and "problems" in meta:
        flop = sum(2.0 * b * r * n * k for b, r, n, k in meta["problems"])
        nbytes = sum(b * (2.0 * r * (n + k) + 4.0 * n * k) for b, r, n, k in meta["problems"])
        return _gemm_both(flop, nbytes)
    if name == "gm3d_attention_qkv_fwd":       # h (B,T,C) + Wqkv (3C,C) in, out (B,T,C)
        B, T, H, C = meta["B"], meta["T"], meta["H"], meta["C"]
        return _gemm_both(B * H * (2.0 * T * 192 * C + 4.0 * T ** 2 * 64), 2.0 * (2 * B * T * C + 3 * C * C))
    return None


def algorithmic(name, meta):
    """-> (bound, amount per launch, unit) for one launch of a hand-written kernel."""
    both = algorithmic_both(name, meta)
    if both is not None:
        return (both["bound"], both["flop"], "FLOP") if both["bound"] == "mfma" else ("hbm", both["bytes"], "B")
    if name == "gm3d_fps":            # 12,288 B xyz + 256 B idx + 768 B centres per cloud
        return "hbm", meta["B"] * (meta["N"] * 12 + meta["npoint"] * 4 + meta["npoint"] * 12), "B"
    if name == "gm3d_knn_group":      # xyz + centres in; idx int64 + neighbourhood + neighbourhood_org out
        g, k = meta["G"], meta["k"]
        return "hbm", meta["B"] * (meta["N"] * 12 + g * 12 + g * k * 8 + 2 * g * k * 12), "B"
    if name == "gm3d_chamfer_fwd":    # 2 clouds in, dist1/dist2 + idx1/idx2 out
        return "hbm", meta["P"] * ((meta["n"] + meta["m"]) * 12 + (meta["n"] + meta["m"]) * 8), "B"
    if name == "gm3d_chamfer_bwd":    # clouds + idx + grads in, 2 gradient clouds out
        return "hbm", meta["P"] * ((meta["n"] + meta["m"]) * (12 + 4 + 4 + 12)), "B"
    sz = 2 if "bfloat16" in str(meta.get("dtype", "")) else 4
    if name == "gm3d_patch_chamfer_loss_fwd":     # per masked patch: 32 predicted + 32 true points in; 2 x 32 neighbour ids + its loss out
        return "hbm", meta["B"] * meta["M"] * (96 * sz + 384 + 256 + 4), "B"
    if name == "gm3d_patch_chamfer_loss_bwd":     # per masked patch: both patches + ids in, 96 gradient values out (the visible rows' zeros not counted)
        return "hbm", meta["B"] * meta["M"] * (96 * sz + 384 + 256 + 96 * sz), "B"
    if name == "gm3d_adamw_ema_flat_step":        # per parameter: P, G, M, V, E read; P, M, V, E written (f32); two bf16 shadows written
        return "hbm", meta["n"] * (9 * 4 + 2 * 2 if meta.get("ema", True) else 7 * 4 + 2), "B"
    if name == "gm3d_attention_fwd":  # QK^T + PV: 4*T^2*64 flop per (b,h)
        return "mfma", meta["B"] * meta["H"] * 4.0 * meta["T"] ** 2 * 64, "FLOP"
    if name == "gm3d_attention_masked_fwd":   # masked attention of the hierarchical encoder: the dense count (blocked pairs are computed too)
        return "mfma", meta["B"] * meta["H"] * 4.0 * meta["T"] ** 2 * meta["HD"], "FLOP"
    if name == "gm3d_attention_masked_bwd":
        return "mfma", meta["B"] * meta["H"] * 10.0 * meta["T"] ** 2 * meta["HD"], "FLOP"
    if name == "gm3d_attention_bwd":  # 5 products (S, dP, dV, dK, dQ): 10*T^2*64 flop per (b,h)
        return "mfma", meta["B"] * meta["H"] * 10.0 * meta["T"] ** 2 * 64, "FLOP"
    if name == "gm3d_residual_ln_fwd":   # res in/out fp32, y + add in, h out
        return "hbm", meta["R"] * 384 * (8 + 3 * sz), "B"
    if name == "gm3d_residual_ln_bwd":   # dh, gin, x in; dx, dy out (+ acc read-modify-write on some calls, not counted)
        return "hbm", meta["R"] * 384 * (12 + 2 * sz), "B"
    if name == "gm3d_add_ln_fwd":        # x, y, z in; s, h out (the any-width residual LayerNorm of the hierarchical model)
        return "hbm", meta["R"] * meta["C"] * 5 * sz, "B"
    if name == "gm3d_add_ln_bwd":        # dh, gin, x in; dx, dy out
        return "hbm", meta["R"] * meta["C"] * 5 * sz, "B"
    if name == "gm3d_bias_gelu_fwd":
        return "hbm", meta["R"] * meta["C"] * 2 * sz, "B"
    if name == "gm3d_bias_gelu_bwd":
        return "hbm", meta["R"] * meta["C"] * 3 * sz, "B"
    if name in ("gm3d_colsum_finish", "gm3d_colsum_finish_batched", "gm3d_colsum_finish_f64"):
        return "hbm", meta["rows"] * meta["cols"] * (8 if name.endswith("f64") else 4), "B"
    # mini-PointNet streaming passes over (G groups, K points, C channels) tiles
    gkc = meta.get("G", 0) * meta.get("K", 0) * meta.get("C", 0) * sz
    passes = {"gm3d_bn_bcast_stats": 1, "gm3d_bn_bcast_apply_relu": 2, "gm3d_bn_bcast_bwd_stats": 2,
              "gm3d_bn_bcast_bwd_apply": 3, "gm3d_group_max_fwd": 1, "gm3d_group_max_bwd": 1, "gm3d_group_scatter_add": 1}
    if name in passes:
        return "hbm", passes[name] * gkc, "B"
    if name == "gm3d_pn_layer1_fwd":
        return "hbm", meta["R"] * (meta["C"] * sz + 12), "B"
    if name == "gm3d_pn_layer1_bwd_stats":
        return "hbm", meta["R"] * (2 * meta["C"] * sz + 12), "B"
    if name == "gm3d_colsum_partial":
        return "hbm", meta["R"] * meta["C"] * sz, "B"
    return None


def roofline_of(name, per_launch, total_ms, dtype="bf16"):
    """One kernel over all its timed launches [(ms, meta), ...] -> {"bound", "achieved", "peak", "unit", "frac", "amount"} (amount =
    algorithmic work per launch in the bound's unit), or None when the kernel has no algorithmic figure.  GEMM-class kernels: flops and
    operand bytes are summed over the launches, the roof that gives the larger time is THE bound, and both rates are reported
    ("mfma": TFLOP/s + frac_mfma, "hbm": GB/s + frac_hbm)."""
    metas = [m for _, m in per_launch]
    if not metas or algorithmic(name, metas[0]) is None:
        return None
    secs = total_ms * 1e-3
    both = [algorithmic_both(name, m) for m in metas]
    if all(b is not None for b in both):
        flop, nbytes = sum(b["flop"] for b in both), sum(b["bytes"] for b in both)
        t_mfma, t_hbm = flop / (MFMA_PEAK_TFLOPS[dtype] * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        out = {"bound": bound, "tflops": round(flop / secs / 1e12, 2), "frac_mfma": round(t_mfma / secs, 4),
               "gbs": round(nbytes / secs / 1e9, 1), "frac_hbm": round(t_hbm / secs, 4)}
        if bound == "mfma":
            out.update(achieved=flop / secs / 1e12, peak=MFMA_PEAK_TFLOPS[dtype], unit="TFLOP/s", amount=flop / len(metas), amount_unit="FLOP")
        else:
            out.update(achieved=nbytes / secs / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", amount=nbytes / len(metas), amount_unit="B")
        out["frac"] = out["achieved"] / out["peak"]
        return out
    bound, _, unit = algorithmic(name, metas[0])
    work = sum(algorithmic(name, m)[1] for m in metas)
    if bound == "hbm":
        out = {"bound": "hbm", "achieved": work / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        out = {"bound": "mfma", "achieved": work / secs / 1e12, "peak": MFMA_PEAK_TFLOPS[dtype], "unit": "TFLOP/s"}
    out.update(frac=out["achieved"] / out["peak"], amount=work / len(metas), amount_unit=unit)
    return out


def _gemm_grid(meta):
    tiles = -(-meta["M"] // 128) * (meta["N"] // 128)
    return (tiles + 7) // 8 * 8 * 256


def pmc_traffic(kernel, dtype, metas=(), pmc_file=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc
    WRITE_SIZE runs of this same bench, eager, summarised in profiles/r02_pmc_fetch_write_per_launch.json), corrected as
    MI355X_MICROARCH.md prescribes for gfx950: counters are in KB, and FETCH_SIZE reports half of a coalesced stream.
    Launch-weighted mean over the kernel's shapes in the step.  None when no measurement is on file."""
    path = os.path.join(ROOT, "profiles", pmc_file or PMC_FILE)
    if not os.path.exists(path):
        return None
    special = {"gm3d_gemm_nt_bf16_multi": "gm3d::gemm_nt_multi_kernel", "gm3d_adamw_ema_flat_step": "gm3d::adamw_ema_flat_kernel",
               "gm3d_gemm_tn_bf16_ws": "gm3d::gemm_tn_ws_kernel", "gm3d_gemm_tn_bf16_ws_pool": "gm3d::gemm_tn_ws_kernel",
               "gm3d_attention_masked_fwd": "gm3d::mattn_fwd_bf16_kernel", "gm3d_attention_masked_bwd": "gm3d::mattn_bwd_bf16_kernel",
               "gm3d_add_ln_fwd": "gm3d::ln_plain_fwd_kernel", "gm3d_add_ln_bwd": "gm3d::ln_plain_bwd_kernel"}
    if kernel in special:
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith(special[kernel])]
    elif kernel == "gm3d_gemm_tn_bf16_ring":
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith("gm3d::gemm_tn_ring_kernel")]
    elif kernel.startswith("gm3d_gemm_tn_bf16_dma"):
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith("gm3d::gemm_tn_dma_kernel")]
    elif kernel == "gm3d_attention_qkv_fwd":
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith("gm3d::attn_qkv_fwd_bf16_kernel")]
    elif kernel == "gm3d_gemm_nt_bf16":
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith("gm3d::gemm_nt_bf16_kernel")]
    elif kernel.startswith("gm3d_gemm_tn_bf16"):
        # the GEMM entry points share one kernel: pick the PMC rows by launch grid (= the timed launches' tile counts)
        grids = {_gemm_grid(m) for m in metas}
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith("gm3d::gemm_tn_bf16_kernel") and r["grid_threads"] in grids]
    else:
        tag = kernel.replace("gm3d_", "gm3d::") + "_kernel"
        rows = [r for r in json.load(open(path)) if r["kernel"].startswith(tag) and (dtype in r["kernel"] or "<" not in r["kernel"])]
    n = sum(r["launches"] for r in rows)
    if not n:
        return None
    return sum(r["launches"] * (2.0 * r["FETCH_SIZE_KB_avg"] + r["WRITE_SIZE_KB_avg"]) * 1024.0 for r in rows) / n


def kernel_sources_sha16():
    """sha256 over gm3d_amd/csrc/* (sorted by name): identifies the kernel code a PMC summary was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gm3d_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_source():
    """Where roofline.traffic comes from and whether the kernels have changed since: the counters are NOT collected in this run
    (rocprofv3 --pmc passes are separate, profiles/README.md)."""
    path = os.path.join(ROOT, "profiles", PMC_SOURCES)
    now = kernel_sources_sha16()
    then = json.load(open(path)).get("sha16") if os.path.exists(path) else None
    return {"file": "profiles/" + PMC_FILE, "kernel_sources_sha16_then": then, "kernel_sources_sha16_now": now, "stale": then != now}


def make_clouds(B, N, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, 3, generator=g) * 2 - 1
    x = x - x.mean(dim=1, keepdim=True)                        # ShapeNet.pc_norm (datasets/ShapeNet55Dataset.py:45-51)
    x = x / x.pow(2).sum(-1).sqrt().amax(dim=1, keepdim=True).unsqueeze(-1)
    return x.to(device).contiguous()


def cpu_baseline(batch, steps):
    """The CPU oracle's pretrain step (fp32, all host threads torch gives us)."""
    from oracle import model_ref as R
    from oracle import ops as oracle_ops
    oracle_ops.build()
    torch.manual_seed(0)
    model = R.PointMAEGM3D().train()
    ema = R.ModelEma(model, 0.999)
    opt = torch.optim.AdamW(R.param_groups(model, 0.05), lr=1e-3)
    x = make_clouds(batch, 1024, 1234, "cpu")
    R.pretrain_step(model, ema, opt, x.clone(), epoch=200, total_epoch=400)       # warm-up
    t0 = time.time()
    for _ in range(steps):
        R.pretrain_step(model, ema, opt, x.clone(), epoch=200, total_epoch=400)
    dt = time.time() - t0
    return {"value": batch * steps / dt, "unit": "clouds/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d pretrain steps of B=%d clouds (N=1024,G=64,k=32,d=384,depth=12) fp32, CPU oracle "
                      "(oracle/model_ref.py + gm3d_oracle.c), %.1f s" % (steps, batch, dt)}


def secondary(device, replays=10):
    """The other configurations of BASELINE.json on the driver's clock: each step captured as a hipGraph and
    `replays` replays timed (after 3 untimed ones) with a synchronise on both sides.  Same kernels, same flat optimizer as the
    headline step.  One GPU, rank 0, bf16.  A leg that fails reports its error instead of a number; none touches the headline
    line's other keys.  -> {name: {clouds_per_s, ms_per_step, batch, workload}}"""
    from gm3d_amd import engine_pretrain as E
    out = {}

    def timed(step, batch):
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(replays):
            o = step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / replays
        v = float(o["loss"])
        if v != v or abs(v) == float("inf"):
            raise FloatingPointError("non-finite loss from the replayed step")
        return {"clouds_per_s": batch / dt, "ms_per_step": dt * 1e3, "batch": batch, "execution": "hipGraph replay"}

    def finetune():            # BASELINE configs[4]: ModelNet40 fine-tune of the pretrained encoder (P/engine_finetune.py:79-183)
        import torch.nn as nn
        from gm3d_amd import engine_finetune as EF
        from gm3d_amd.point_transformer import PointTransformer
        B = 32
        torch.manual_seed(0)
        model = PointTransformer(dict(trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64,
                                      encoder_dims=384)).to(device).train()
        crit = nn.CrossEntropyLoss()
        opt = EF.build_optimizer(model, lr=5e-4, flat=True, max_norm=10.0)
        fa = SimpleNamespace(lr=5e-4, min_lr=1e-6, warmup_epochs=10, epochs=300)
        pool = [make_clouds(B, 8192, 100 + i, device) for i in range(3)]
        targets = (torch.arange(B, device=device) * 7) % 40
        g = EF.GraphedFinetuneStep(model, crit, opt, pool[0], targets, npoints=1024, max_norm=10.0, bf16=True, overlap_sampling=True)

        def step(i):
            EF.adjust_learning_rate(opt, 20 + i / 100.0, fa)
            return g(pool[i % 3], targets, next_points=pool[(i + 1) % 3])
        r = timed(step, B)
        r["workload"] = "ModelNet40 fine-tune step: B=32 clouds of 8192 points, FPS 8192->1200 + 1024-subset, G=64, k=32, cls 40"
        return r

    def published():           # the published-run variant (P/engine_pretrain_Classifier_SVM.py:40-332)
        from gm3d_amd import engine_pretrain_Classifier_SVM as EV
        from gm3d_amd import models_mae_learn_loss_Classifier_SVM_feature_besed as V
        from gm3d_amd.point_mae import Point_MAE
        B = 128
        torch.manual_seed(0)
        model = V.mae_vit_base_patch16_dec512d8b().to(device).train()
        teacher = Point_MAE({"group_size": 32, "num_group": 64, "loss": "cdl2",
                             "transformer_config": {"mask_ratio": 0, "mask_type": "rand", "trans_dim": 384, "encoder_dims": 384,
                                                    "depth": 12, "drop_path_rate": 0.1, "num_heads": 6, "decoder_depth": 4,
                                                    "decoder_num_heads": 6}}).to(device).eval()
        for p in teacher.parameters():
            p.requires_grad_(False)
        ema = E.ModelEma(model, E.ema_decay_for_epoch(150))
        opt = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=ema)
        pa = SimpleNamespace(mask_ratio=0.6, epochs=300, relative=True, bf16=True, accum_iter=1, after_epoch=15,
                             loss_multiply_by=(13.889, 1000.0), after_200_epoch=False, shared_learnable_tokens=False, lr=1e-3, min_lr=0.0,
                             warmup_epochs=10)
        pool = [make_clouds(B, 1024, 500 + i, device) for i in range(3)]
        g = EV.graphed_step(model, ema, teacher, opt, pa, pool[0], 150)
        r = timed(lambda i: g(pool[i % 3]), B)
        r["workload"] = ("published-run pretrain step (EMA teacher + student with 12-block loss-prediction decoder + frozen Point-MAE "
                         "teacher): B=128, N=1024, G=64, k=32")
        return r

    def m2ae():                # BASELINE configs[3]: Point-M2AE + GM3D (Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99)
        from gm3d_amd import point_m2ae as P
        B = 128
        torch.manual_seed(0)
        model = P.PointM2AE().to(device).train()
        ema = E.ModelEma(model, 0.999)
        opt = E.build_optimizer(model, lr=1e-3, flat=True, model_ema=ema)
        ma = SimpleNamespace(bf16=True, epochs=300)
        pool = [make_clouds(B, 2048, 100 + i, device) for i in range(3)]
        # the whole step as ONE graph (point_m2ae.GraphedM2AEStep -- the next batch's grouping on a second stream -- measures the same:
        # profiles/NEGATIVE_RESULTS.md, round 4)
        for i in range(2):
            P.pretrain_step(model, ema, opt, pool[i].clone(), 100, ma)
        torch.cuda.synchronize()
        static_in = pool[0].clone()
        from gm3d_amd import streams
        g = torch.cuda.CUDAGraph()
        with streams.capture(g):
            res = P.pretrain_step(model, ema, opt, static_in, 100, ma)

        def step(i):
            static_in.copy_(pool[i % 3])
            g.replay()
            return res
        r = timed(step, B)
        r["workload"] = "Point-M2AE+GM3D pretrain step: B=128 clouds of 2048 points, G=512/256/64, k=16/8/8, dims 96/192/384"
        # roofline of this step's dominant hand-written kernel (HIP-event brackets over two eager steps right after the replays; the
        # counters behind `traffic` are the committed --pmc passes over tools/bench_m2ae.py)
        from gm3d_amd import ops as O
        probe = O.KernelTimer()
        O.set_kernel_timer(probe)
        try:
            for i in range(2):
                P.pretrain_step(model, ema, opt, pool[i].clone(), 100, ma)
        finally:
            O.set_kernel_timer(None)
        ps = probe.summary()
        cand = [n for n in ps if algorithmic(n, ps[n]["meta"])]
        if cand:
            dom = max(cand, key=lambda n: ps[n]["total_ms"])
            rf = roofline_of(dom, ps[dom]["per_launch"], ps[dom]["total_ms"])
            r["roofline"] = {"kernel": dom, "bound": rf["bound"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"],
                             "frac": rf["frac"], "traffic": pmc_traffic(dom, "bf16", [m for _, m in ps[dom]["per_launch"]], PMC_FILE_M2AE),
                             "avg_launch_us": ps[dom]["avg_ms"] * 1e3, "launches_per_step": ps[dom]["launches"] / 2,
                             "ms_per_step": ps[dom]["total_ms"] / 2, "algorithmic_per_launch": rf["amount"], "algorithmic_unit": rf["amount_unit"],
                             "both_bounds": {k: rf[k] for k in ("tflops", "frac_mfma", "gbs", "frac_hbm") if k in rf} or None}
            top = sorted(cand, key=lambda n: -ps[n]["total_ms"])[:8]
            r["kernel_rooflines"] = {}
            for n in top:
                q = roofline_of(n, ps[n]["per_launch"], ps[n]["total_ms"])
                r["kernel_rooflines"][n] = {"bound": q["bound"], "frac": round(q["frac"], 4), "achieved": round(q["achieved"], 2), "unit": q["unit"],
                                            "ms_per_step": round(ps[n]["total_ms"] / 2, 3), "launches_per_step": ps[n]["launches"] / 2}
        return r

    for name, fn in (("finetune_modelnet", finetune), ("published_run", published), ("point_m2ae", m2ae)):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as ex:           # a secondary leg never takes the headline line down
            out[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


import contextlib  # noqa: E402


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL and gloo print banners on fd 1 when their first communicator comes up: keep stdout for the ONE JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, dry_run=False):
    """`python bench.py --gpus N` without a launcher's environment: start N fresh rank processes of this file (one per GPU;
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them), relay rank 0's ONE JSON line, exit with
    the worst return code.  The parent never initialises HIP (device_count() does not, on this image).  What the reference does
    with torch.distributed.launch / SLURM variables (P/main_pretrain_multi_gpu.py:166-177,309-311, P/util/misc.py:215-247).
    A rank that dies takes the others down with it (exact PIDs) instead of leaving them in the rendezvous."""
    import subprocess
    have = torch.cuda.device_count()
    if not dry_run and n > have:
        print("bench.py: --gpus %d but only %d GPU(s) visible: refusing to measure fewer ranks than asked for" % (n, have), file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stdout is ours (the JSON line); the other ranks print nothing on stdout by contract: route it to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    worst, live, ours = 0, set(range(n)), set()          # ours: ranks this launcher terminated (their -15 is not a result)
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and r not in ours:
                print("bench.py: rank %d exited with %d" % (r, rc), file=sys.stderr)
                worst = max(worst, rc if rc > 0 else 128 - rc)
                for o in live:
                    ours.add(o)
                    procs[o].terminate()
        time.sleep(0.05)
    return worst


def dry_run_rank(rank, world):
    """--launch-dry-run: the rank environment without a GPU -- gloo group, all-reduce of the rank numbers, one JSON line."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("GM3D_DRYRUN_FAIL_RANK") == str(rank):      # test hook: a rank that dies before the rendezvous
        sys.exit(7)
    with stdout_to_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        n = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": n, "rank_sum": float(t), "backend": "gloo"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="clouds per GPU (BASELINE config: 128)")
    ap.add_argument("--fp32", action="store_true", help="parity precision instead of bf16")
    ap.add_argument("--epoch", type=int, default=200, help="epoch index (200/400: guided-mask branch active)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (fine-tune, published run, Point-M2AE)")
    ap.add_argument("--cpu-batch", type=int, default=16)     # 1 warm-up + 3 timed steps of B=16: ~25 s of CPU work on the box's host cores
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--bucket-mb", type=int, default=256,
                    help="gradient all-reduce chunk; the default covers the whole 147 MB flat gradient buffer in ONE collective "
                         "(graph mode cannot overlap it with backward anyway, and one large ring all-reduce has the least fixed cost)")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager launches instead of hipGraph replay (the step is ~1000 launches: eager is host-bound)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="start the ranks exactly as a real run would, but each only joins a gloo group, all-reduces its rank "
                         "number and exits (no GPU, no kernels): the CPU test of the launch path")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="NOT a measurement: N ranks over gloo, all of them on GPU 0 -- runs the whole N > 1 code path (rank launch, "
                         "segmented step, collectives, aggregation of the line) on a one-GPU box; the line carries \"rehearsal\"")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher above us: be the launcher (before any GPU call)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], dry_run=args.launch_dry_run or args.rehearse_gloo))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        # never measure a different number of ranks than the line would claim
        sys.exit("bench.py: --gpus %d disagrees with WORLD_SIZE %d" % (args.gpus, world))
    if args.launch_dry_run:
        return dry_run_rank(rank, world)
    if args.rehearse_gloo:
        local_rank = 0                      # every rank on GPU 0 (RCCL refuses that: gloo carries the collectives)
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: LOCAL_RANK %d but %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
    # GM3D_FORCE_DIST=1: take the data-parallel code path (RCCL process group, bucketed all-reduce, two graphs) with a
    # single rank -- lets a one-GPU box rehearse exactly what the N>1 launch runs
    use_dist = world > 1 or os.environ.get("GM3D_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        with stdout_to_stderr():
            if args.rehearse_gloo:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if use_dist else 0)

    from gm3d_amd import engine_pretrain as E
    from gm3d_amd import models_mae_learn_loss as M
    from gm3d_amd import ops

    # no TunableOp table: since round 3 every product of the bf16 step runs on a hand-written kernel (gemm.choose never names the
    # library), so the measured path does not depend on hipBLASLt solution choices
    tuned = False
    torch.manual_seed(0)                      # identical random-init weights on every rank
    model = M.mae_vit_base_patch16_dec512d8b(norm_pix_loss=False).to(device).train()
    model_ema = E.ModelEma(model, decay=E.ema_decay_for_epoch(args.epoch))
    use_graph = not args.no_graph
    # data-parallel: lay the flat gradient buffer out by backward segment so that each segment is one all-reduce range
    segmented = use_dist and not args.no_graph
    optimizer = E.build_optimizer(model, lr=1e-3, weight_decay=0.05, flat=True, model_ema=model_ema,
                                  segment_of=E.ddp_segment if segmented else None)
    grad_sync = E.GradSync.from_flat(optimizer, bucket_bytes=args.bucket_mb << 20) if use_dist else None
    step_args = SimpleNamespace(mask_ratio=0.6, epochs=400, relative=True, bf16=not args.fp32, accum_iter=1,
                                lr=1e-3, min_lr=0.0, warmup_epochs=40)
    torch.manual_seed(1234 + rank)            # per-rank augmentation / mask / DropPath streams
    pool = [make_clouds(args.batch, 1024, 1234 + rank + 1000 * i, device) for i in range(4)]

    def eager_step(i):
        E.adjust_learning_rate(optimizer, args.epoch + i / 1000.0, step_args)
        return E.pretrain_step(model, model_ema, optimizer, pool[i % len(pool)].clone(), args.epoch, step_args,
                               grad_sync=grad_sync)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Eager probe: every hand-written kernel bracketed by HIP events -> which one dominates, and per-kernel time.
    eager_step(0)                 # un-timed: first launches pay code-object loading
    probe = ops.KernelTimer()
    ops.set_kernel_timer(probe)
    nprobe = 2 if use_graph else max(args.warmup, 1)
    for i in range(nprobe):
        out = eager_step(i)
    ops.set_kernel_timer(None)
    if os.environ.get("GM3D_BENCH_DEBUG"):
        print("after eager probe:", {k: float(v) for k, v in out.items() if v.numel() == 1}, file=sys.stderr)
    psum = probe.summary()
    dominant = max((n for n in psum if algorithmic(n, psum[n]["meta"])), key=lambda n: psum[n]["total_ms"])

    graph_note = None
    seg_note = None
    if use_graph and segmented:
        # four graphs with the all-reduce of each backward segment issued while the next segment runs (SegmentedDDPStep);
        # any problem falls back to the two-graph layout below
        try:
            graphed = E.SegmentedDDPStep(model, model_ema, optimizer, step_args, pool[0], args.epoch)
        except Exception as ex:
            seg_note = "segmented capture failed (%s: %s)" % (type(ex).__name__, str(ex)[:120])
            segmented = False
    else:
        segmented = False
    if use_dist and use_graph:    # every rank must take the same layout
        flag = torch.tensor([1.0 if segmented else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        segmented = segmented and float(flag) == 1.0
    if use_graph and not segmented:
        try:
            graphed = E.GraphedPretrainStep(model, model_ema, optimizer, step_args, pool[0], args.epoch, grad_sync=grad_sync)
        except Exception as ex:  # never lose the measurement to a capture problem: fall back to eager launches
            graph_note = "capture failed (%s: %s); eager" % (type(ex).__name__, str(ex)[:120])
            use_graph = False
            if grad_sync is not None:
                grad_sync.overlap = True
    if use_dist:                  # every rank must take the same path
        flag = torch.tensor([1.0 if use_graph else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if use_graph and float(flag) == 0.0:
            use_graph, graph_note = False, "another rank failed to capture; eager"
            grad_sync.overlap = True
    if use_graph:
        def step(i):
            E.adjust_learning_rate(optimizer, args.epoch + i / 1000.0, step_args)
            return graphed(pool[i % len(pool)])
        for i in range(args.warmup):
            out = step(i)
        timer = None
    else:
        step = eager_step
        timer = ops.KernelTimer(only=[dominant])
        for i in range(args.warmup if graph_note else 0):
            out = step(i)

    fence()
    ops.set_kernel_timer(timer)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step spread: events between the steps,
    t0 = time.perf_counter()                                                          # no synchronisation inside the timed region
    for i in range(args.steps):
        marks[i].record()
        out = step(args.warmup + i)
        if os.environ.get("GM3D_BENCH_DEBUG"):
            print("step", i, {k: float(v) for k, v in out.items() if v.numel() == 1}, "lr", [float(g["lr"]) for g in optimizer.param_groups],
                  "in", float(graphed.static_in.abs().mean()) if use_graph else None, file=sys.stderr)
    marks[args.steps].record()
    fence()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if timer is None:
        # a captured graph cannot carry per-kernel events: re-run the same steps eagerly, right after the timed
        # region, with the dominant kernel bracketed by HIP events on its launch stream
        timer = ops.KernelTimer(only=[dominant])
        ops.set_kernel_timer(timer)
        for i in range(min(args.steps, 5)):
            eager_step(i)
        ops.set_kernel_timer(None)
        roofline_timing = "HIP events in an eager re-run of %d steps right after the timed hipGraph-replay region" % min(args.steps, 5)
    else:
        roofline_timing = "HIP events inside the timed region"
    loss = float(out["loss"] + out["loss_learn"])
    assert loss == loss and abs(loss) != float("inf"), "non-finite loss in the timed region"

    if rank == 0:
        dtype = "f32" if args.fp32 else "bf16"
        tsum = timer.summary()[dominant]
        # No overhead is subtracted: an event bracket adds ~2-3 us to a ~9 us kernel (rocprof's duration for the same
        # kernel is in profiles/), so `achieved` here is the conservative figure; the empty-bracket time is reported.
        overhead_ms = ops.KernelTimer.bracket_overhead_ms()
        raw_avg_ms = tsum["avg_ms"]
        # the kernel runs on several shapes per step (e.g. 8192- and 3200-row token streams): algorithmic work and time
        # are summed over the timed launches, so `achieved` is total work / total kernel time
        dom = roofline_of(dominant, tsum["per_launch"], tsum["total_ms"], dtype)
        bound, achieved, peak, runit, amount, unit = dom["bound"], dom["achieved"], dom["peak"], dom["unit"], dom["amount"], dom["amount_unit"]
        # every hand-written kernel against its own roofline (eager probe steps, raw event brackets); GEMM-class kernels carry both
        # rates (tflops / frac_mfma, gbs / frac_hbm) and `bound` names the roof that binds
        all_roof = {}
        for n, v in psum.items():
            r_ = roofline_of(n, v["per_launch"], v["total_ms"], dtype)
            if r_ is None:
                continue
            all_roof[n] = {k: (round(x, 4) if isinstance(x, float) else x) for k, x in r_.items() if k not in ("amount", "amount_unit", "peak")}
        for v in psum.values():
            v.pop("per_launch", None)
        per_step = {n: {"launches_per_step": v["launches"] / nprobe,
                        "avg_us": round(v["avg_ms"] * 1e3, 2),
                        "ms_per_step": round(v["total_ms"] / nprobe, 4)} for n, v in psum.items()}
        line = {
            "metric": "point-clouds/sec pretrain step (N=1024,G=64)",
            "value": args.batch * world * args.steps / dt,
            "unit": "clouds/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if use_dist and not args.rehearse_gloo else None,   # the RCCL group's own count (None: no group, one GPU)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_spread": {"min": round(step_ms[0], 4), "median": round(step_ms[len(step_ms) // 2], 4),
                                   "max": round(step_ms[-1], 4), "how": "HIP events between consecutive steps of the timed region (rank 0)"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "Point-MAE+GM3D pretrain step, B=%d clouds/GPU, N=1024, G=64, k=32, d=384, "
                                   "depth=12 (+2x4 decoder blocks), mask_ratio 0.6, epoch %d/400 (guided mask), "
                                   "random-init weights" % (args.batch, args.epoch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "roofline": {"kernel": dominant, "bound": bound, "achieved": achieved, "peak": peak, "unit": runit,
                         "frac": achieved / peak, "traffic": pmc_traffic(dominant, dtype, [m for _, m in tsum["per_launch"]]),
                         "avg_launch_us": raw_avg_ms * 1e3, "empty_bracket_us": overhead_ms * 1e3, "launches_timed": tsum["launches"],
                         "algorithmic_per_launch": amount, "algorithmic_unit": unit, "timing": roofline_timing,
                         "both_bounds": {k: dom[k] for k in ("tflops", "frac_mfma", "gbs", "frac_hbm") if k in dom} or None,
                         "traffic_source": pmc_source()},
            "execution": ("hipGraph replay" + ((" (4 graphs: segment all-reduces overlap the next backward segment)" if segmented else
                                             " (fwd+bwd | all-reduce | update)") if use_dist else "")) if use_graph
            else (graph_note or "eager"),
            "execution_note": seg_note,
            "kernel_rooflines": all_roof,
            "tuned_gemm_table": bool(tuned),
            "hip_kernels_ms_per_step": per_step,
            "loss": loss,
        }
        if args.rehearse_gloo:
            line["rehearsal"] = ("%d gloo ranks sharing GPU 0: the N > 1 code path end to end, NOT a measurement "
                                 "(value is meaningless)" % dist.get_world_size())
        if not args.no_secondary and not use_dist and not args.fp32:
            line["secondary"] = secondary(device)
        # N=1 only: at N>1 the other ranks would sit in a GPU barrier while rank 0 does 20 s of host work
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps)
        elif world > 1:
            line["cpu_baseline"] = None
            line["cpu_baseline_note"] = "measured at N=1 only (rank 0 of the 1-GPU run)"
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
