"""CPU oracle of the Point-M2AE hierarchical grouping.  TEST INFRASTRUCTURE ONLY.
No reference source exists for this model (Point-M2AE_SA3D/README.md:1); this restates the configuration
(cfgs/config_Point_M2AE.yaml:57-99) on the oracle's FPS / KNN: "parity unpinned"."""
import torch

from . import ops


def hierarchical_group(pts, num_groups=(512, 256, 64), group_sizes=(16, 8, 8)):
    nbs, centers, idxs = [], [], []
    src = pts
    for G, k in zip(num_groups, group_sizes):
        B, N, _ = src.shape
        fidx = ops.furthest_point_sample(src, G)
        center = ops.gather_operation(src.transpose(1, 2).contiguous(), fidx).transpose(1, 2).contiguous()
        _, idx = ops.KNN(k=k, transpose_mode=True)(src, center)
        flat = (idx + torch.arange(0, B).view(-1, 1, 1) * N).view(-1)
        nb = src.reshape(B * N, -1)[flat, :].view(B, G, k, 3) - center.unsqueeze(2)
        nbs.append(nb); centers.append(center); idxs.append(idx)
        src = center
    return nbs, centers, idxs


def local_attention_mask(center, radius):
    d = (center.unsqueeze(2) - center.unsqueeze(1)).pow(2).sum(-1).sqrt()
    return d >= radius


def propagate_visibility(mask_coarse, idxs):
    masks = [mask_coarse]
    for lvl in range(len(idxs) - 1, 0, -1):
        B, G_prev = idxs[lvl].shape[0], idxs[lvl - 1].shape[1]
        vis_prev = torch.zeros(B, G_prev, dtype=torch.bool)
        for b in range(B):
            for g in range(idxs[lvl].shape[1]):
                if not masks[0][b, g]:
                    vis_prev[b, idxs[lvl][b, g]] = True
        masks.insert(0, ~vis_prev)
    return masks
