"""The EMA teacher's inference pass as two half-batch chains on two streams (fused.NOGRAD_SPLIT = 2) and as one chain (= 1), eager, repeated
beside a loader process: which of the two differs from its quiet result, and in which output?   python tools/split_stress.py [rounds]"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
from gm3d_amd import fused, models_mae_learn_loss as M
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
model = M.mae_vit_base_patch16_dec512d8b().cuda().eval()
x = torch.randn(64, 1024, 3, device="cuda") * 0.3
mask = torch.zeros(64, 64, dtype=torch.bool, device="cuda")


def run(ns):
    fused.NOGRAD_SPLIT = ns
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(x, mask=mask, num_visible=64, need_pix_pred=False)
    torch.cuda.synchronize()
    return {k: o[k].float().clone() for k in ("loss_pred", "features", "center", "neighborhood")}


quiet = {ns: run(ns) for ns in (2, 1)}
print("quiet: split == whole:", {k: torch.equal(quiet[2][k], quiet[1][k]) for k in quiet[2]})
env = dict(os.environ, READY_FILE="/tmp/split_ready", STOP_FILE="/tmp/split_stop")
for f in (env["READY_FILE"], env["STOP_FILE"]):
    if os.path.exists(f):
        os.remove(f)
child = None
if os.environ.get("LOAD", "1") == "1":
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "replay_stress.py"), "--load", "64", "300"], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL)
    while not os.path.exists(env["READY_FILE"]):
        time.sleep(0.5)
bad = {ns: {k: 0 for k in quiet[2]} for ns in (2, 1)}
first = {}
for i in range(N):
    for ns in (2, 1):
        r = run(ns)
        for k in r:
            if not torch.equal(r[k], quiet[ns][k]):
                bad[ns][k] += 1
                if (ns, k) not in first:
                    d = (r[k] != quiet[ns][k])
                    idx = d.nonzero()
                    first[(ns, k)] = (i, int(d.sum()), tuple(r[k].shape), idx[0].tolist(), idx[-1].tolist(), float((r[k] - quiet[ns][k]).abs().max()))
print("%d rounds: outputs that differed from the quiet run" % N)
for ns in (2, 1):
    print("  NOGRAD_SPLIT=%d: %s" % (ns, bad[ns]))
for k, v in first.items():
    print("  first", k, v)
if child is not None:
    open(env["STOP_FILE"], "w").write("x")
    print(child.communicate(timeout=120)[0].decode().strip())
