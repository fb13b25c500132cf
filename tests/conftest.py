import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DDP_RESULTS = {"dir": None, "rc": None, "note": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """The two-rank test of the real model (tests/test_gpu_ddp.py) needs two FRESH processes sharing cuda:0.  They are started
    here -- after collection, before any test (i.e. before this pytest process has made a single HIP call: a process that has
    initialised the GPU must neither exec nor be the template of GPU children) -- run to completion, and leave result files
    the test reads.  Only when that test is actually selected."""
    if not any("test_gpu_ddp" in item.nodeid for item in session.items):
        return
    import torch
    if torch.cuda.device_count() < 1:        # counting devices does not initialise the GPU on this image
        DDP_RESULTS["note"] = "no GPU"
        return
    out = tempfile.mkdtemp(prefix="gm3d_ddp_")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # the invariant the docstring states, checked: nothing imported during collection may have touched the GPU
    assert not torch.cuda.is_initialized(), "a test module initialised HIP at import: the DDP workers need a fresh parent"
    procs = []
    for rank in range(2):
        log = open(os.path.join(out, "rank%d.log" % rank), "w")
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.ddp_worker", "--rank", str(rank), "--world", "2", "--port",
                                       str(port), "--out", out], cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT))
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=900))
        except subprocess.TimeoutExpired:
            p.kill()                        # the exact child we started
            rcs.append(-9)
    DDP_RESULTS.update(dir=out, rc=rcs)


@pytest.fixture(scope="session")
def ddp_results():
    return DDP_RESULTS


@pytest.fixture(scope="session")
def oracle_ops():
    """The CPU oracle (test infrastructure only)."""
    from oracle import ops
    ops.build()
    return ops
