"""-m gpu: BASELINE config #3's code path with the REAL model and TWO ranks (P/main_pretrain_multi_gpu.py:309-311 is the DDP wrap
it replaces).  Two fresh processes share cuda:0 over gloo (tests/conftest.py starts them before this process touches the GPU;
tests/ddp_worker.py is one rank), B=4 clouds per rank, SegmentedDDPStep eager and hipGraph-captured, bf16.

Checked: (1) ranks start from different seeds and the constructor's broadcast makes them equal; (2) the flat gradient buffer
after the three collectives of a step is identical on both ranks and equals the mean of the two single-rank segment gradients
(per-rank BatchNorm batch statistics, as the reference's DDP implies: no SyncBN); (3) parameters and the EMA teacher's
parameters stay BIT-identical across ranks after 3 steps; (4) BatchNorm running statistics differ per rank after a step and
rank 0's win at the broadcast; (5) the captured layout reproduces the eager layout's trajectory."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(ddp_results, mode):
    d = ddp_results["dir"]
    if d is None:
        pytest.fail("the two-rank workers were not started: %s" % ddp_results["note"])
    logs = ""
    for r in range(2):
        for name in ("error_rank%d.txt" % r, "rank%d.log" % r):
            f = os.path.join(d, name)
            if os.path.exists(f):
                logs += "\n--- %s ---\n%s" % (name, open(f).read()[-3000:])
    assert ddp_results["rc"] == [0, 0], "worker exit codes %s%s" % (ddp_results["rc"], logs)
    return [torch.load(os.path.join(d, "%s_rank%d.pt" % (mode, r)), weights_only=True) for r in range(2)]


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_two_rank_segmented_step_real_model(ddp_results, mode):
    r0, r1 = _load(ddp_results, mode)
    # (1) same start although the seeds differed
    assert torch.equal(r0["p_start"], r1["p_start"])
    # (2) averaged gradient: identical on both ranks, and the mean of the two local ones
    assert torch.equal(r0["g_avg"], r1["g_avg"])
    assert not torch.equal(r0["g_local"], r1["g_local"])
    # g_local is ALWAYS computed eagerly (also in graph mode: tests/ddp_worker.py), g_avg by the step under test.  The step's
    # kernels are deterministic and a captured launch runs the same code on the same operands, gloo sums the two ranks' buffers
    # (one rounding) and the step halves the sum (exact): the averaged gradient EQUALS (g0 + g1) / 2 computed here.
    want = (r0["g_local"] + r1["g_local"]) / 2
    for seg, (lo, hi) in r0["segments"].items():
        a, b = r0["g_avg"][lo:hi], want[lo:hi]
        if not torch.equal(a, b):      # name the parameters (diagnosis of a failure)
            offs = r0["offs"] + [r0["g_avg"].numel()]
            worst = sorted(((float((r0["g_avg"][o:e] - want[o:e]).abs().max()) / float(b.abs().max()), n)
                            for n, o, e in zip(r0["names"], offs[:-1], offs[1:]) if lo <= o < hi), reverse=True)[:6]
            loc = [float((r["g_avg"][lo:hi] - r["g_local"][lo:hi]).abs().max()) for r in (r0, r1)]
            raise AssertionError((mode, seg, worst, loc))
    # (3) replicas stay bit-identical
    assert torch.equal(r0["params"], r1["params"]) and not torch.equal(r0["params"], r0["p_start"])
    assert torch.equal(r0["ema"], r1["ema"])
    # (4) per-rank BatchNorm statistics, rank 0's win at the broadcast
    assert not torch.equal(r0["buf_pre"], r1["buf_pre"])
    assert torch.equal(r1["buf_post"], r0["buf_pre"]) and torch.equal(r0["buf_post"], r0["buf_pre"])
    for a, b in zip(r0["losses"], r1["losses"]):
        assert all(x == x for x in a + b)
        assert abs(a[2] - b[2]) <= 1e-6 * abs(a[2])          # the gradient norm is of the averaged gradient: same on both ranks


def test_two_rank_graph_equals_eager(ddp_results):
    e0, _ = _load(ddp_results, "eager")
    g0, _ = _load(ddp_results, "graph")
    assert e0["losses"] == g0["losses"], (e0["losses"], g0["losses"])          # exact, step by step
    assert torch.equal(e0["params"], g0["params"]) and torch.equal(e0["ema"], g0["ema"])
    assert torch.equal(e0["g_avg"], g0["g_avg"])


def test_two_rank_segmented_step_at_the_full_per_rank_batch(ddp_results):
    """BASELINE config #3's per-rank size (B = 128 clouds per rank) through the captured four-graph step with two ranks (gloo, one GPU):
    the averaged gradient EQUALS the mean of the two eagerly computed shard gradients, and the replicas' parameters and EMA teacher are
    bit-identical after 3 steps (compared inside the workers / by SHA-256: the flat buffers are 147 MB each)."""
    f0, f1 = _load(ddp_results, "full")
    assert f0["batch_per_rank"] == 128 and f0["shards_differ"]
    assert f0["avg_equals_mean"] and f1["avg_equals_mean"], [(f["max_dev"], f["n_differ"], f["segments_equal"], f["worst"]) for f in (f0, f1)]
    assert f0["g_avg_hash"] == f1["g_avg_hash"]
    assert f0["params_hash"] == f1["params_hash"] and f0["ema_hash"] == f1["ema_hash"]
    for a, b in zip(f0["losses"], f1["losses"]):
        assert all(x == x and abs(x) != float("inf") for x in a + b)
        assert abs(a[2] - b[2]) <= 1e-6 * abs(a[2])


def test_two_rank_train_one_epoch(ddp_results):
    """The drop-in epoch loop (engine_pretrain.train_one_epoch, P/engine_pretrain.py:38-271) with a process group of two ranks: it builds
    its own gradient synchroniser (which broadcasts rank 0's state: the ranks were seeded differently), runs the first iterations
    eagerly and the rest as replays of the captured four-graph step.  Replicas must end bit-identical."""
    e0, e1 = _load(ddp_results, "epoch")
    assert torch.equal(e0["params"], e1["params"]) and torch.equal(e0["ema"], e1["ema"])
    for st in (e0["stats"], e1["stats"]):
        assert st["replayed_iters"] == 7 - 3 and st["capture_s"] > 0
        assert all(v == v and abs(v) != float("inf") for v in st.values())
    assert abs(e0["stats"]["loss"] - e1["stats"]["loss"]) <= 1e-6 * abs(e0["stats"]["loss"])      # the epoch's metric all-reduce


def test_bench_n2_code_path_end_to_end_over_gloo():
    """`python bench.py --gpus 2 --rehearse-gloo`: the launcher starts two ranks, both on this GPU, gloo carries the collectives -- everything
    else is the N > 1 bench path the driver's 8-GPU node runs (state broadcast, eager probe through GradSync, the captured four-graph step
    with its three collectives per step, the agreement all-reduces, MAX over ranks of the timed region, rank 0's line).  Not a measurement:
    the line says so."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-gloo", "--batch", "32", "--steps", "3",
                        "--warmup", "2", "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] is None and "rehearsal" in d and d["rehearsal"].startswith("2 gloo ranks")
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert "4 graphs" in d["execution"], d["execution"]
    assert d["loss"] == d["loss"] and d["value"] > 0 and d["cpu_baseline"] is None
