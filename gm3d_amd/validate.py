"""Linear-SVM validation of the pretrained encoder: the reference's only quality signal during pretraining.

Mirror of P/main_pretrain_multi_gpu.py::validate / evaluate_svm (:414-483), P/utils/miscc.py::fps (:13-20) and
P/utils/dist_utils.py::gather_tensor (:50-54), SURVEY.md 8(f).1:
    points (B,8192,3) --FPS--> (B,1024,3) --model(points, all-False mask, noaug=True)--> (B,64,384)
    pooled = mean over tokens + max over tokens  -> all ranks gathered -> sklearn SVC(C=0.01, kernel='linear') on the CPU.
Differences of execution only: FPS emits the sampled points in the same launch (no separate gather), and the token
features are pooled on the GPU BEFORE the all-gather (the SVM never sees the un-pooled tokens), which cuts the
collective from 64x384 to 384 floats per cloud.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops


def fps(data, number):
    """miscc.fps: data (B,N,3) -> the `number` farthest-point samples (B,number,3), in sampling order."""
    return ops.fps(data.contiguous().float(), number)[1]


@torch.no_grad()
def extract_features(model, points, npoints=1024, bf16=False):
    """-> token features (B,64,384) of the eval-mode encoder (P/:427-433)."""
    raw = model.module if hasattr(model, "module") else model
    pts = fps(points, npoints)      # the reference resamples unconditionally (P/main_pretrain_multi_gpu.py:430)
    assert pts.size(1) == npoints
    mask = torch.zeros(pts.shape[0], raw.num_group, dtype=torch.bool, device=pts.device)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        return raw(pts, mask, noaug=True, num_visible=raw.num_group).float()


def pool_features(features):
    """(B,T,C) -> (B,C): mean(1) + max(1), the SVM input of evaluate_svm (:479,481)."""
    return features.mean(1) + features.max(1)[0]


def gather_tensor(tensor, world_size=None):
    """dist_utils.gather_tensor: all_gather + concat along dim 0 (identity without a process group)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tensor
    outs = [torch.empty_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, tensor.contiguous())
    return torch.cat(outs, dim=0)


def evaluate_svm(train_features, train_labels, test_features, test_labels):
    """P/:476-483.  Accepts token features (n,T,C) like the reference or already pooled (n,C)."""
    from sklearn.svm import SVC
    def pooled(f):
        f = np.asarray(f)
        return f.mean(1) + f.max(1) if f.ndim == 3 else f
    clf = SVC(C=0.01, kernel="linear")
    clf.fit(pooled(train_features), np.asarray(train_labels))
    pred = clf.predict(pooled(test_features))
    return np.sum(np.asarray(test_labels) == pred) * 1.0 / pred.shape[0]


@torch.no_grad()
def validate(model, extra_train_dataloader, test_dataloader, npoints=1024, device="cuda", bf16=False):
    """Loaders yield (taxonomy_ids, model_ids, (points, label)) like the reference's ModelNet loaders (:428-431).
    Returns the SVM accuracy (computed on every rank from the gathered features, like the reference)."""
    was_training = model.training
    model.eval()
    feats = {}
    for split, loader in (("train", extra_train_dataloader), ("test", test_dataloader)):
        f, l = [], []
        for _, _, data in loader:
            points, label = data[0].to(device), data[1].to(device)
            f.append(pool_features(extract_features(model, points, npoints, bf16=bf16)))
            l.append(label.view(-1))
        feats[split] = (gather_tensor(torch.cat(f, 0)), gather_tensor(torch.cat(l, 0)))
    model.train(was_training)
    return evaluate_svm(feats["train"][0].cpu().numpy(), feats["train"][1].cpu().numpy(),
                        feats["test"][0].cpu().numpy(), feats["test"][1].cpu().numpy())
