"""`python bench.py --gpus N` starts N ranks by itself or fails loudly.  CPU only: --launch-dry-run makes
each child join a gloo group, all-reduce its rank number and exit before anything of gm3d_amd is imported.
Reference behaviour being mirrored: P/main_pretrain_multi_gpu.py:166-177,309-311 + P/util/misc.py:215-247 (one process per
GPU, rank / world size from the environment)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)
    return p, time.time() - t0


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    p, _ = _run(["--gpus", "2", "--launch-dry-run"])
    assert p.returncode == 0, p.stderr
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and len(p.stdout.strip().splitlines()) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2
    assert lines[0]["rank_sum"] == 1.0            # ranks 0 and 1 both took part in the all-reduce


def test_gpus_4_rank_sum():
    p, _ = _run(["--gpus", "4", "--launch-dry-run"])
    assert p.returncode == 0, p.stderr
    (line,) = _json_lines(p.stdout)
    assert line["rccl_ranks"] == 4 and line["rank_sum"] == 6.0


def test_single_rank_needs_no_launcher():
    p, _ = _run(["--launch-dry-run"])
    assert p.returncode == 0, p.stderr
    (line,) = _json_lines(p.stdout)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1


def test_more_gpus_than_visible_fails_loudly():
    import torch
    n = torch.cuda.device_count() + 1
    p, _ = _run(["--gpus", str(max(n, 2))])
    assert p.returncode != 0
    assert "refusing" in p.stderr and not _json_lines(p.stdout)


def test_gpus_disagreeing_with_world_size_fails_loudly():
    p, _ = _run(["--gpus", "4", "--launch-dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert "disagrees with WORLD_SIZE" in p.stderr and not _json_lines(p.stdout)
    p, _ = _run(["--gpus", "1", "--launch-dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not _json_lines(p.stdout)


def test_under_an_external_launcher_the_environment_is_used():
    """the driver's N>1 form: WORLD_SIZE / RANK come from torch.distributed.run; bench.py must not start ranks of its own"""
    port = "29611"
    env = {"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port}
    base = {k: v for k, v in os.environ.items()}
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--launch-dry-run"],
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r), **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert len(_json_lines(outs[0][0])) == 1 and not _json_lines(outs[1][0])


def test_a_dead_rank_takes_the_job_down():
    p, dt = _run(["--gpus", "2", "--launch-dry-run"], {"GM3D_DRYRUN_FAIL_RANK": "1"})
    assert p.returncode == 7, (p.returncode, p.stderr)
    assert not _json_lines(p.stdout)
    assert dt < 120          # rank 0 is terminated instead of waiting out the rendezvous timeout
