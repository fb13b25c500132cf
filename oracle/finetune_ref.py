"""CPU oracle of the fine-tune / classification path: plain PyTorch fp32 + oracle/ops.py.

TEST INFRASTRUCTURE ONLY (same rule as oracle/model_ref.py: imported by tests/ only, never by gm3d_amd/).
Restates P/models/Point_MAE.py:444-579 (PointTransformer), P/engine_finetune.py:108-151 (one iteration) and
P/util/lr_decay.py (layer-wise lr decay groups).  P/ = /root/reference/Point-MAE_SA3D/.
Pinned against the reference's own models/Point_MAE.py by tests/golden/make_golden_finetune.py (this container
only) -> tests/golden/finetune_b4.npz, checked in tests/test_oracle_golden.py.  FPS / KNN stay "parity unpinned"
(oracle/gm3d_oracle.c header).
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import model_ref as R

POINT_ALL = {1024: 1200, 2048: 2400, 4096: 4800, 8192: 8192}          # P/engine_finetune.py:117-126


class PointTransformer(nn.Module):
    def __init__(self, trans_dim=384, depth=12, drop_path_rate=0.1, cls_dim=40, num_heads=6, group_size=32, num_group=64,
                 encoder_dims=384):
        super().__init__()
        self.group_divider = R.Group(num_group, group_size)
        self.encoder = R.Encoder(encoder_dims)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, trans_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, trans_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, trans_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = R.TransformerEncoder(trans_dim, depth, num_heads, dpr)
        self.norm_p = nn.LayerNorm(trans_dim)
        self.cls_head_finetune = nn.Sequential(                                   # P/models/Point_MAE.py:482-492
            nn.Linear(trans_dim * 2, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, cls_dim))
        self.loss_ce = nn.CrossEntropyLoss()

    def forward(self, pts):                                                       # :551-575
        neighborhood, center, _ = self.group_divider(pts)
        tokens = self.encoder(neighborhood)
        B = tokens.size(0)
        x = torch.cat((self.cls_token.expand(B, -1, -1), tokens), dim=1)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1), self.pos_embed(center)), dim=1)
        x = self.norm_p(self.blocks(x, pos))
        return self.cls_head_finetune(torch.cat([x[:, 0], x[:, 1:].max(1)[0]], dim=-1))

    def get_loss_acc(self, ret, gt):                                              # :499-503
        loss = self.loss_ce(ret, gt.long())
        acc = (ret.argmax(-1) == gt).sum() / float(gt.size(0))
        return loss, acc * 100


def layer_id(name, num_layers=12):                                                # P/util/lr_decay.py:65-78
    if name in ("cls_token", "pos_embed") or name.startswith("patch_embed"):
        return 0
    if name.startswith("blocks"):
        return int(name.split(".")[2]) + 1
    return num_layers


def param_groups_lrd(model, weight_decay=0.05, layer_decay=0.75, num_layers=12):  # :15-62, no_weight_decay_list never matches
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    groups = {}
    for n, p in model.named_parameters():
        nd = p.ndim == 1
        lid = layer_id(n, num_layers)
        g = groups.setdefault((lid, nd), {"lr_scale": scales[lid], "weight_decay": 0.0 if nd else weight_decay, "params": []})
        g["params"].append(p)
    return list(groups.values())


def sample_points(points, npoints, subset):                                       # P/engine_finetune.py:117-134
    point_all = min(POINT_ALL[npoints], points.size(1))
    idx = ops.furthest_point_sample(points, point_all)
    idx = idx[:, torch.as_tensor(np.asarray(subset), dtype=torch.long)].contiguous()
    return ops.gather_operation(points.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()


def finetune_step(model, optimizer, points, targets, npoints, subset, scale, shift, max_norm=None):
    """fp32 (the reference's AMP is a precision choice, not part of the algorithm).  -> loss, outputs, grad norm."""
    pts = R.scale_and_translate_(sample_points(points, npoints, subset), scale, shift)
    outputs = model(pts)
    loss = nn.functional.cross_entropy(outputs, targets.long())
    optimizer.zero_grad()
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) if max_norm is not None else None
    optimizer.step()
    return loss.detach(), outputs.detach(), gnorm
