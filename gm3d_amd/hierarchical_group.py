"""Hierarchical multi-scale grouping for the Point-M2AE configuration (BASELINE config #4):
three nested FPS + KNN levels -- N=2048 points -> 512 groups of 16 -> 256 groups of 8 (over the 512 centres) -> 64 groups of 8
(over the 256 centres) -- plus the two pieces of bookkeeping the multi-scale masked auto-encoder hangs on them: the
local-radius attention masks (0.32 / 0.64 / 1.28) and the back-projection of the coarsest level's visibility mask to the finer
levels.  Hyper-parameters: Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99.

The reference ships NO source for this model (Point-M2AE_SA3D/README.md:1: "will be released soon"), so this
module follows the configuration and the published Point-M2AE description (Zhang et al., NeurIPS 2022, sec. 3.1-3.2) and its
parity is against our own oracle only ("parity unpinned").  Each level is one FPS launch and one fused KNN+gather+centre launch of
the same kernels the Point-MAE path uses (gm3d_fps, gm3d_knn_group).
"""
import torch
import torch.nn as nn

from . import ops


class HierarchicalGroup(nn.Module):
    def __init__(self, num_groups=(512, 256, 64), group_sizes=(16, 8, 8)):
        super().__init__()
        assert len(num_groups) == len(group_sizes)
        self.num_groups, self.group_sizes = tuple(num_groups), tuple(group_sizes)

    def forward(self, pts):
        """pts (B,N,3) f32 -> per level lists: neighbourhoods (B,G_l,k_l,3) centred on their centre, centres (B,G_l,3),
        idx (B,G_l,k_l) int64 into the PREVIOUS level's points (level 0: the raw cloud; level l>0: level l-1's centres)."""
        neighborhoods, centers, idxs = [], [], []
        src = pts.contiguous()
        for G, k in zip(self.num_groups, self.group_sizes):
            center = ops.fps(src, G)[1]
            nb, _, idx = ops.knn_group(src, center, k, return_idx=True)
            neighborhoods.append(nb)
            centers.append(center)
            idxs.append(idx)
            src = center
        return neighborhoods, centers, idxs


def local_attention_mask(center, radius):
    """(B,G,3) -> (B,G,G) bool, True where attention is NOT allowed: centres farther apart than `radius`."""
    return torch.cdist(center, center) >= radius


def propagate_visibility(mask_coarse, idxs):
    """mask_coarse (B,G_last) bool, True = masked, at the coarsest level -> per-level masks [(B,G_0), .., (B,G_last)]:
    a finer group is visible iff it is a member of at least one visible group of the next coarser level."""
    masks = [mask_coarse]
    for lvl in range(len(idxs) - 1, 0, -1):
        vis_coarse = ~masks[0]                                         # (B,G_l)
        idx = idxs[lvl]                                                # (B,G_l,k_l) -> members among level l-1 groups
        B, G_prev = idx.shape[0], idxs[lvl - 1].shape[1]
        vis_prev = torch.zeros(B, G_prev, dtype=torch.bool, device=idx.device)
        sel = idx.masked_select(vis_coarse.unsqueeze(-1).expand_as(idx))   # data-dependent size: host sync, bookkeeping only
        rows = torch.arange(B, device=idx.device).view(B, 1, 1).expand_as(idx).masked_select(vis_coarse.unsqueeze(-1).expand_as(idx))
        vis_prev[rows, sel] = True
        masks.insert(0, ~vis_prev)
    return masks
