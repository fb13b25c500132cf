"""One EAGER forward + backward of the north-star model (the model's own forward / forward_loss / forward_learning_loss, plain .backward():
what the fixture tests run), repeated on the same inputs beside a loader process; every parameter gradient compared bit for bit with the
quiet run.    python tools/eager_stress.py [B] [rounds] [fp32|bf16]"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
from gm3d_amd import models_mae_learn_loss as M
from tests import clouds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
DP = os.environ.get("DROPPATH") == "1"           # DropPath left on, its draws re-seeded before every run
FRESH = os.environ.get("FRESH") == "1"           # a NEW model (same seed) for every run: first-use paths (caches keyed by weight, lazy state)


def build():
    torch.manual_seed(3)
    m = M.mae_vit_base_patch16_dec512d8b().cuda().train()
    if not DP:
        for mod in m.modules():
            if isinstance(mod, M.DropPath):
                mod.drop_prob = 0.0
    return m


model = build()
x = clouds.gaussian(B, 1024, 5).cuda()
mask = torch.zeros(B, 64, dtype=torch.bool)
mask[:, torch.randperm(64, generator=torch.Generator().manual_seed(1))[:39]] = True
mask = mask.cuda()


def run():
    global model
    if FRESH:
        model = build()
    torch.manual_seed(77)
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        s = model(x, mask=mask)
        Mn = s["mask_num"]
        lo = model.forward_loss(s["pix_pred"][:, -Mn:], s["neighborhood"], s["mask"])
        ll = model.forward_learning_loss(s["loss_pred"][:, -Mn:], mask, lo["matrix"].detach(), relative=True)
    (lo["Chamfer_mean"] + ll).backward()
    torch.cuda.synchronize()
    out = {"fwd::" + k: s[k].detach().float().clone() for k in ("features", "pix_pred", "loss_pred")}
    out["fwd::chamfer"] = lo["Chamfer_mean"].detach().clone()
    out["fwd::learn"] = ll.detach().clone()
    out.update({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    return out


run()
quiet = run()
again = run()
unstable = [k for k in quiet if not torch.equal(quiet[k], again[k])]
print("quiet repeat: %d of %d tensors differ%s" % (len(unstable), len(quiet), (" " + str(unstable[:6])) if unstable else ""), flush=True)
env = dict(os.environ, READY_FILE="/tmp/eager_ready", STOP_FILE="/tmp/eager_stop")
for f in (env["READY_FILE"], env["STOP_FILE"]):
    if os.path.exists(f):
        os.remove(f)
child = None
if os.environ.get("LOAD", "1") == "1":
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "replay_stress.py"), "--load", "64", "600"], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL)
    while not os.path.exists(env["READY_FILE"]):
        time.sleep(0.5)
bad_rounds, per = 0, {}
for i in range(N):
    r = run()
    diff = [k for k in quiet if k not in unstable and not torch.equal(r[k], quiet[k])]
    if diff:
        bad_rounds += 1
        for k in diff:
            per.setdefault(k, [0, 0.0])
            per[k][0] += 1
            per[k][1] = max(per[k][1], float((r[k].float() - quiet[k].float()).abs().max() / quiet[k].float().abs().max().clamp_min(1e-30)))
        if bad_rounds <= 3:
            print("  round %d: %d tensors differ, first few: %s" % (i, len(diff), diff[:8]), flush=True)
print("%s B=%d: %d of %d rounds differed from the quiet run" % ("bf16" if bf16 else "fp32", B, bad_rounds, N))
for k, (c, m) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
    print("   %4d x  max rel %.2e  %s" % (c, m, k))
if child is not None:
    open(env["STOP_FILE"], "w").write("x")
    print(child.communicate(timeout=120)[0].decode().strip())
