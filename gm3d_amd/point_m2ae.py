"""Point-M2AE + GeoMask3D: the hierarchical multi-scale masked auto-encoder of BASELINE config #4 on the
MI355X-native operators.

The reference ships NO source for this model -- Point-M2AE_SA3D/README.md:1 ("will be released soon"); only the hyper-parameters
exist (Point-M2AE_SA3D/cfgs/config_Point_M2AE.yaml:57-99: N=2048; groups 512x16 / 256x8 / 64x8; encoder dims 96/192/384 with
5 blocks each, 6 heads, local radii 0.32/0.64/1.28; decoder dims 384/192 with 1 block each, 1 up-block each; mask ratio 0.8).
The model below follows those numbers and the published Point-M2AE design (Zhang et al., NeurIPS 2022: multi-scale masking by
back-projecting the coarsest visibility, token embedding of the previous level's features, local-radius attention, hierarchical
decoder with 3-NN token propagation, Chamfer reconstruction of the masked level-1 patches), with GeoMask3D's teacher / student
scoring (P/engine_pretrain.py:86-171) put on the COARSEST tokens.  Parity is against our own restatement only
(oracle/hier_ref.py): "parity unpinned".  Choices the configuration does not fix, stated once:

  * attention between tokens i, j of a level is allowed iff both are visible AND their centres are closer than the level's radius;
  * every level keeps ALL its tokens in place (static shapes; a masked token is excluded by the attention mask, computes garbage
    that is discarded, and hands its un-encoded embedding to the next level -- the same information flow as compacting the
    visible tokens, padding to the batch maximum and scattering back);
  * GeoMask3D: the loss predictor (same head as P/models_mae_learn_loss.py:152-158,668,677) reads the coarsest decoder stage; a
    coarse token's target is the mean Chamfer loss of its masked level-1 members; the guided mask is generate_mask of the
    north-star model (P/:744-784) over the 64 coarsest tokens with this configuration's mask ratio.

Kernels: gm3d_fps / gm3d_knn_group per level, gm3d_knn (3-NN propagation), gm3d_attention_masked_fwd/bwd (T up to 512, head_dim
16/32/64, bitset mask), gm3d_attention_* (unmasked 64-token stage), gm3d_chamfer_*, gm3d_mask_select, gm3d_rank_loss.  The
level-0 token embed runs on embed.EmbedFn, every Linear on heads.LinearBiasFn; LayerNorm and the deeper levels' BatchNorm are
PyTorch modules.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import models_mae_learn_loss as M
from . import ops
from .hierarchical_group import HierarchicalGroup


FUSED_EMBED0 = True
FUSED_LAYERNORM = True
FUSED_POS = True          # positional MLPs through heads.PosEmbedFn (tests flip it)
FUSED_BLOCKS = True       # block stacks: residual sums inside the LayerNorm kernel, bias + GELU as one pass, biases of proj / fc2 added
#                           (and their gradients summed) by the LayerNorm kernels -- heads.AddLayerNormFn / BiasGeluFn; any width
ACT_TAPS = None           # tests set this to a list: the sign pattern (pre-activation > 0) of every ReLU / LeakyReLU of a grad-enabled
#                           forward on the per-op path, in call order (six token-embed ReLUs, the loss head's LeakyReLU, two
#                           token-propagation ReLUs)
POOL_TAPS = None          # tests set this to a list: every max-pool of a grad-enabled TokenEmbed.forward (per-op path) appends its
#                           winner weights (groups, k, C): 1 at the maximum, 1/n at each of n exactly tied maxima -- how
#                           torch.amax's backward shares the gradient; the discrete decisions a gradient comparison has to share
#                           (tests/test_gpu_m2ae.py)
FUSED_EMBED_DEEP = True   # levels 1-2 token embeds: BatchNorm on the streaming kernels, no concatenation (bf16 mode)
FUSED_PROPAGATION = True  # token propagation: 3-NN interpolation + concatenation as one launch (heads.Interp3Fn), deterministic backward
DEFER_WGRADS = True       # pretrain_step: weight gradients of the layers outside the block stacks as one launch at the end of backward
STACK_NODE = True         # a block stack as ONE autograd node (masked_stack.MaskedStackFn): the weight gradients of all its blocks in one
#                           launch, one column-sum finish per kind, one transposing launch; FUSED_BLOCKS' per-op nodes are the cross-check
VISIBLE_FIRST = True      # student pass: every level's stack runs on the visible tokens moved to the front of the cloud, cut to the
#                           static bound the mask generator implies (12 of 64 -> 16, 96 of 256, all 512): the attention kernels skip the
#                           filler tiles, levels 1-2 shrink to 3/8 and 1/4 of their rows.  Same results for visible tokens.


def radius_mask(center, radius):
    """(B,G,3) -> (B,G,G) bool, True = centres at distance >= radius (no attention).  Squared distances accumulated coordinate by
    coordinate in fp32 (separately rounded products and sums: the CPU oracle reproduces every bit)."""
    d = center.unsqueeze(2) - center.unsqueeze(1)
    d2 = d[..., 0] * d[..., 0]
    d2 = d2 + d[..., 1] * d[..., 1]
    d2 = d2 + d[..., 2] * d[..., 2]
    r = torch.tensor(radius, dtype=torch.float32)
    return d2 >= float(r * r)            # radius^2 as an fp32 product, like the kernel (gm3d_radius_mask_bits)


def back_project(mask_coarse, idxs):
    """Multi-scale masking: mask_coarse (B,G_last) bool (True = masked) -> [mask_0 .. mask_last]; a finer group is visible iff at
    least one visible group of the next coarser level contains it.  Static shapes, no host sync: one launch per level on the GPU
    (gm3d_back_project), an integer scatter-add elsewhere."""
    masks = [mask_coarse]
    if mask_coarse.is_cuda and mask_coarse.dtype == torch.bool:
        from ._capi import lib
        for lvl in range(len(idxs) - 1, 0, -1):
            idx = idxs[lvl].contiguous()
            B, Gc, k = idx.shape
            Gf = idxs[lvl - 1].shape[1]
            out = torch.empty(B, Gf, dtype=torch.bool, device=idx.device)
            ops._launch("gm3d_back_project", {"B": B, "Gc": Gc, "Gf": Gf}, lib.gm3d_back_project, ops._ptr(masks[0].contiguous()),
                        ops._ptr(idx), B, Gc, k, Gf, ops._ptr(out), ops._stream())
            masks.insert(0, out)
        return masks
    for lvl in range(len(idxs) - 1, 0, -1):
        idx = idxs[lvl]                                               # (B,G_l,k_l) members among the level l-1 groups
        B, G_prev = idx.shape[0], idxs[lvl - 1].shape[1]
        vis = (~masks[0]).unsqueeze(-1).expand_as(idx).reshape(B, -1).to(torch.int32)
        cnt = torch.zeros(B, G_prev, dtype=torch.int32, device=idx.device).scatter_add_(1, idx.reshape(B, -1), vis)
        masks.insert(0, cnt == 0)
    return masks


def _linear(x, weight, bias=None):
    """x @ weight^T + bias for an nn.Linear / Conv1d(k=1) weight.  On the GPU through heads.LinearBiasFn: bf16 weight shadows
    instead of autocast's per-call casts, the weight gradient in fp32, and the bias gradient by our own column-sum kernel -- the
    PyTorch reduction behind nn.Linear's bias gradient comes back wrong from hipGraph replays at B = 128 (8192+ rows x 768+
    columns; tools/m2ae_step_diag.py)."""
    if x.is_cuda:
        from . import heads
        return heads.LinearBiasFn.apply(x, weight, bias, heads._adt())
    return F.linear(x, weight.reshape(weight.shape[0], -1), bias)


class Linear(nn.Linear):
    def forward(self, x):
        return _linear(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm on our own row kernel (heads.LayerNormFn) on the GPU: widths 96 / 192 / 384, output in the activation type."""

    def forward(self, x):
        from . import heads
        if FUSED_LAYERNORM and self.elementwise_affine and heads.layer_norm_supported(x, x.shape[-1]):
            return heads.LayerNormFn.apply(x, self.weight, self.bias, self.eps, heads._adt())
        return super().forward(x)


class Mlp(nn.Module):
    """timm Mlp (same attribute names as models_mae_learn_loss.Mlp: the state-dict keys do not change)."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def _lin(x, conv, bn=None, act=False):
    """Conv1d(k=1) (+ BatchNorm1d + ReLU) on a (rows, C) layout."""
    if bn is not None and act and FUSED_EMBED_DEEP and x.is_cuda and x.shape[0] % 8 == 0:
        from . import heads
        C = conv.out_channels
        if heads._adt() == torch.bfloat16 and heads.bn_bcast_supported(x, C, 8):
            # Conv + BatchNorm + ReLU on the embed's streaming kernels: the conv bias is the (constant) per-group term
            t = heads.ExpandRowsFn.apply(conv.bias.view(1, 1, -1), 1, x.shape[0] // 8, torch.bfloat16).reshape(-1, C)
            return heads.bn_bcast_act(_linear(x, conv.weight, None), t, bn, 8)
    y = _linear(x, conv.weight, conv.bias)
    if bn is not None:
        y = bn(y)
    if act and ACT_TAPS is not None and torch.is_grad_enabled():
        ACT_TAPS.append(y.detach() > 0)
    return F.relu(y) if act else y


class TokenEmbed(nn.Module):
    """mini-PointNet of one level: (B,G,k,in_c) -> (B,G,out_c).  Level 0 reads xyz (the Point-MAE embed's layer sizes), deeper
    levels read the previous level's token features."""

    def __init__(self, in_c, out_c):
        super().__init__()
        a, b = (128, 256) if in_c == 3 else (in_c, in_c)
        mid = 512 if in_c == 3 else out_c
        self.first_conv = nn.Sequential(nn.Conv1d(in_c, a, 1), nn.BatchNorm1d(a), nn.ReLU(inplace=True), nn.Conv1d(a, b, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(2 * b, mid, 1), nn.BatchNorm1d(mid), nn.ReLU(inplace=True), nn.Conv1d(mid, out_c, 1))
        self.out_c = out_c

    def _forward_fused(self, groups):
        """The same layers with the BatchNorms on the embed's streaming kernels (heads.BnBcastActFn) and the concatenation
        [global | local] replaced by the two halves of the second conv's weight (local rows + a per-group term): no (rows, 2b)
        tensor, no PyTorch BatchNorm over 262,144+ rows."""
        from . import heads
        B, G, k, C = groups.shape
        adt = heads._adt()
        c0, bn0, _, c1 = self.first_conv
        c2, bn1, _, c3 = self.second_conv
        x = groups.reshape(B * G * k, C)
        t0 = heads.ExpandRowsFn.apply(c0.bias.view(1, 1, -1), 1, B * G, adt).reshape(B * G, -1)
        a1 = heads.bn_bcast_act(_linear(x, c0.weight, None), t0, bn0, k)
        f = _linear(a1, c1.weight, c1.bias)                                            # (rows, b)
        b_ = f.shape[-1]
        pool = (lambda t: heads.GroupMaxFn.apply(t)) if heads.group_max_supported(f.view(B * G, k, b_)) else (lambda t: t.amax(dim=1))
        fg = pool(f.view(B * G, k, b_))                                                # (B*G, b)
        W3 = c2.weight.squeeze(-1)                                                     # (mid, 2b): [global | local] columns
        a2 = heads.bn_bcast_act(_linear(f, W3[:, b_:], None), _linear(fg, W3[:, :b_], c2.bias), bn1, k)
        z = _linear(a2, c3.weight, c3.bias)
        return pool(z.view(B * G, k, self.out_c)).view(B, G, self.out_c)

    def forward(self, groups):
        B, G, k, C = groups.shape
        if FUSED_EMBED_DEEP and groups.is_cuda and C != 3:
            from . import heads
            if heads._adt() == torch.bfloat16 and all(heads.bn_bcast_supported(groups, c, k) for c in
                                                      (self.first_conv[0].out_channels, self.second_conv[0].out_channels)):
                return self._forward_fused(groups)
        x = groups.reshape(B * G * k, C)
        f = _lin(_lin(x, self.first_conv[0], self.first_conv[1], True), self.first_conv[3])                  # (rows, b)
        fg = f.view(B * G, k, -1).amax(dim=1, keepdim=True).expand(-1, k, -1)
        y = torch.cat([fg, f.view(B * G, k, -1)], dim=-1).reshape(B * G * k, -1)
        y = _lin(_lin(y, self.second_conv[0], self.second_conv[1], True), self.second_conv[3])
        if POOL_TAPS is not None and torch.is_grad_enabled():
            for t in (f.detach().view(B * G, k, -1), y.detach().view(B * G, k, self.out_c)):
                win = (t == t.amax(dim=1, keepdim=True)).to(t.dtype)
                POOL_TAPS.append(win / win.sum(dim=1, keepdim=True))
        return y.view(B * G, k, self.out_c).amax(dim=1).view(B, G, self.out_c)


class MaskedAttention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=False)
        self.proj = Linear(dim, dim)

    def forward(self, x, bits, proj_bias=True):
        qkv = self.qkv(x)
        hd = x.shape[-1] // self.num_heads
        if bits is None and hd == 64 and x.shape[1] <= 128:
            a = ops.attention(qkv, self.num_heads, self.scale)              # the Point-MAE kernel (whole head in one workgroup)
        else:
            a = ops.attention_masked(qkv, bits, self.num_heads, self.scale)
        return self.proj(a) if proj_bias else _linear(a, self.proj.weight, None)


class MaskedBlock(nn.Module):
    def __init__(self, dim, num_heads, drop_path=0.0):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn = MaskedAttention(dim, num_heads)
        self.drop_path = M.DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, 4 * dim)

    def forward(self, x, bits=None):
        x = x + self.drop_path(self.attn(self.norm1(x), bits))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class BlockStack(nn.Module):
    def __init__(self, dim, depth, num_heads, dpr):
        super().__init__()
        self.blocks = nn.ModuleList([MaskedBlock(dim, num_heads, dpr[i]) for i in range(depth)])

    def forward(self, x, pos, bits=None):
        if STACK_NODE and FUSED_BLOCKS and x.is_cuda:
            from . import heads, masked_stack
            if masked_stack.supported(x, self.blocks):
                return masked_stack.run_stack(self.blocks, x, pos, bits, self.training, heads._adt())
        if FUSED_BLOCKS and x.is_cuda:
            from . import heads
            if heads.layer_norm_supported(x, x.shape[-1]) and x.shape[-1] % 8 == 0:
                return self._forward_fused(x, pos, bits)
        for blk in self.blocks:                 # the position is re-added before every block, like Point-MAE
            x = blk(x + pos, bits)
        return x

    def _forward_fused(self, x, pos, bits):
        """The same blocks with every residual sum formed inside the following LayerNorm's kernel:
            s, h = AddLN(s, pending branch output (+ its Linear's bias, x its DropPath factor), pos)     block input + norm1
            s, h = AddLN(s, proj output (+ proj.bias, x DropPath factor))                                  norm2
            pending = fc2(GELU(fc1(h) + fc1.bias))   -- bias and GELU one pass; fc2's bias joins the next AddLN
        15 of the ~20 elementwise launches per block (forward + backward) and the separate bias-gradient sums disappear."""
        from . import heads
        adt = heads._adt()
        B = x.shape[0]
        s, y, yb, rs = x, None, None, None
        # the DropPath factors of the whole stack (two sites per block) from ONE uniform draw + one launch, not four launches per site
        probs = []
        for blk in self.blocks:
            p = blk.drop_path.drop_prob if isinstance(blk.drop_path, M.DropPath) else 0.0
            probs += [p, p]
        scales = M.drop_path_scales(B, probs, self.training, x.device)
        for i, blk in enumerate(self.blocks):
            s, h = heads.AddLayerNormFn.apply(s, y, yb, rs, pos, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, adt)
            a = blk.attn(h, bits, proj_bias=False)
            s, h = heads.AddLayerNormFn.apply(s, a, blk.attn.proj.bias, scales[2 * i], None, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, adt)
            g = heads.BiasGeluFn.apply(_linear(h, blk.mlp.fc1.weight, None), blk.mlp.fc1.bias, adt)
            y, yb = _linear(g, blk.mlp.fc2.weight, None), blk.mlp.fc2.bias
            rs = scales[2 * i + 1]
        return heads.ResidualTailFn.apply(s, y, yb, rs, adt)


class TokenPropagation(nn.Module):
    """Up-block of the hierarchical decoder: the coarse level's tokens interpolated to the fine centres (3 nearest coarse centres,
    inverse squared-distance weights), concatenated with the fine level's own tokens, then a shared MLP."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        last = in_channel
        for c in mlp:
            self.mlp_convs.append(nn.Conv1d(last, c, 1))
            self.mlp_bns.append(nn.BatchNorm1d(c))
            last = c

    @staticmethod
    def neighbours(xyz_fine, xyz_coarse):
        """-> (idx (B,N,3) int64, w (B,N,3) f32): the three nearest coarse centres of every fine centre and their normalised
        inverse-squared-distance weights.  A function of the grouping alone: teacher and student of one step share it."""
        with torch.no_grad():
            dist, idx = ops.knn(xyz_coarse, xyz_fine, 3)                   # (B,N,3): Euclidean distances, ascending
            w = 1.0 / (dist * dist + 1e-8)
            w = w / w.sum(dim=-1, keepdim=True)
        return idx, w

    def forward(self, xyz_fine, xyz_coarse, tok_fine, tok_coarse, nbrs=None):
        B, N, _ = xyz_fine.shape
        idx, w = nbrs if nbrs is not None else self.neighbours(xyz_fine, xyz_coarse)
        from . import heads
        if FUSED_PROPAGATION and tok_fine.is_cuda and tok_fine.dtype in ops._DT and tok_fine.shape[-1] % 8 == 0 and tok_coarse.shape[-1] % 8 == 0:
            y = heads.Interp3Fn.apply(tok_fine, tok_coarse, idx, w).reshape(B * N, -1)       # interpolation + concatenation: one launch
        else:
            near = heads.take_rows(tok_coarse, idx.reshape(B, N * 3)).view(B, N, 3, -1)     # (deterministic backward: a coarse token has many readers)
            interp = (near * w.unsqueeze(-1).to(near.dtype)).sum(dim=2)
            y = torch.cat([tok_fine, interp.to(tok_fine.dtype)], dim=-1).reshape(B * N, -1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            y = _lin(y, conv, bn, True)
        return y.view(B, N, -1)


def _pos_mlp(dim):
    return nn.Sequential(Linear(3, dim), nn.GELU(), Linear(dim, dim))


def _pos(mlp, centers):
    """a positional MLP Linear(3,C) -> GELU -> Linear(C,C) on (B,G,3) centres.  GPU: the north-star model's fused node
    (heads.PosEmbedFn: the K = 3 layer + GELU as one streaming kernel, its backward a pure reduction, the C x C layer on our GEMM)."""
    if FUSED_POS and centers.is_cuda and centers.dim() == 3 and mlp[0].out_features % 8 == 0:
        from . import heads
        return heads.PosEmbedFn.apply(centers, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, heads._adt())
    return mlp(centers)


class PointM2AE(nn.Module):
    """config: the `model` section of cfgs/config_Point_M2AE.yaml (dict or attribute object)."""

    def __init__(self, config=None):
        super().__init__()
        c = dict(mask_ratio=0.8, group_sizes=[16, 8, 8], num_groups=[512, 256, 64], encoder_depths=[5, 5, 5],
                 encoder_dims=[96, 192, 384], local_radius=[0.32, 0.64, 1.28], decoder_depths=[1, 1], decoder_dims=[384, 192],
                 decoder_up_blocks=[1, 1], drop_path_rate=0.1, num_heads=6)
        if config is not None:
            c.update({k: (config[k] if isinstance(config, dict) else getattr(config, k)) for k in c
                      if (k in config if isinstance(config, dict) else hasattr(config, k))})
        self.cfg = c
        self.mask_ratio, self.num_heads = c["mask_ratio"], c["num_heads"]
        self.num_group = c["num_groups"][-1]                                # the tokens GeoMask3D scores and masks
        dims, depths = c["encoder_dims"], c["encoder_depths"]
        assert len(dims) == 3 and c["decoder_dims"] == [dims[2], dims[1]], "built for the three-scale configuration"
        self.local_radius = list(c["local_radius"])
        self.group_divider = HierarchicalGroup(c["num_groups"], c["group_sizes"])
        self.token_embed = nn.ModuleList([TokenEmbed(3 if i == 0 else dims[i - 1], dims[i]) for i in range(3)])
        self.encoder_pos_embeds = nn.ModuleList([_pos_mlp(d) for d in dims])
        dpr = [x.item() for x in torch.linspace(0, c["drop_path_rate"], sum(depths))]
        self.encoder_blocks, at = nn.ModuleList(), 0
        for d, n in zip(dims, depths):
            self.encoder_blocks.append(BlockStack(d, n, self.num_heads, dpr[at:at + n]))
            at += n
        self.encoder_norms = nn.ModuleList([LayerNorm(d) for d in dims])
        ddims, ddepths = c["decoder_dims"], c["decoder_depths"]
        ddpr = [x.item() for x in torch.linspace(0, c["drop_path_rate"], sum(ddepths))]
        self.mask_token = nn.Parameter(torch.zeros(1, 1, ddims[0]))
        nn.init.trunc_normal_(self.mask_token, std=0.02)
        self.decoder_pos_embeds = nn.ModuleList([_pos_mlp(d) for d in ddims])
        self.h_decoder = nn.ModuleList([BlockStack(ddims[0], ddepths[0], self.num_heads, ddpr[:ddepths[0]]),
                                        BlockStack(ddims[1], ddepths[1], self.num_heads, ddpr[ddepths[0]:])])
        self.token_prop = nn.ModuleList([TokenPropagation(ddims[0] + ddims[1], [ddims[1] * 4, ddims[1]])])
        self.decoder_norm = LayerNorm(ddims[1])
        self.rec_head = nn.Conv1d(ddims[1], 3 * c["group_sizes"][1], 1)
        self.loss_pred_head = nn.Sequential(nn.Conv1d(ddims[0], 1024, 1), nn.BatchNorm1d(1024), nn.LeakyReLU(negative_slope=0.2),
                                            nn.Conv1d(1024, ddims[0], 1))
        self.loss_func = ops.ChamferDistanceL2()

    # ------------------------------------------------------------------ forward
    def _compact_bounds(self, vis_count):
        """static row counts of the visible-first order per level, from the number of visible COARSEST tokens per cloud (None:
        unknown -> every level keeps all its rows and is only re-ordered)."""
        G, k = self.cfg["num_groups"], self.cfg["group_sizes"]
        if vis_count is None:
            return list(G)
        up = lambda n, m: (n + m - 1) // m * m
        b2 = min(G[2], up(vis_count, 16))
        b1 = min(G[1], up(vis_count * k[2], 32))
        b0 = min(G[0], up(vis_count * k[2] * k[1], 32))
        return [b0, b1, b2]

    def encode(self, neighborhoods, centers, idxs, masks, vis_count=None, compact=False):
        """-> per level: encoder outputs for every token position (only the visible ones are meaningful).
        compact: run each level's stack in the visible-first order (masked_stack.partition_visible)."""
        outs, prev = [], None
        bounds = self._compact_bounds(vis_count) if compact else None
        for i in range(3):
            if i == 0:
                from . import heads
                if neighborhoods[0].is_cuda and FUSED_EMBED0 and heads._adt() == torch.bfloat16:
                    # the level-0 embed has Point-MAE's layer structure (xyz in, 16-point groups): the north-star model's fused
                    # mini-PointNet (embed.EmbedFn: analytic layer-1 statistics, BatchNorm folded into streaming passes, no
                    # concatenation, fp32 weight gradients) -- 1 M rows here, the largest tensors of the step: 53.5 -> 40.1 ms.
                    # Throughput (bf16) mode only: in fp32 the per-op modules stay, which is the path tests/test_gpu_m2ae.py pins
                    # against the oracle (the fused node's fp32 BatchNorm sums are two-stage fp32 -- within 3e-5 of the modules,
                    # tests/test_gpu_embed.py, but the oracle test's fp64 criterion for weights in front of a BatchNorm is tighter)
                    from . import embed
                    tok = embed.run_embed(self.token_embed[0], neighborhoods[0])
                else:
                    tok = self.token_embed[0](neighborhoods[0])
            else:
                B, G, k = idxs[i].shape
                from . import heads
                tok = self.token_embed[i](heads.take_rows(prev, idxs[i].reshape(B, G * k)).view(B, G, k, -1))
            if compact:
                from . import masked_stack as S
                with torch.no_grad():
                    part = S.partition_visible(masks[i], bounds[i])
                    cen_c = S.select_rows(centers[i], part["perm_c"])
                    bits = ops.radius_mask_bits(cen_c, part["vis_c"], self.local_radius[i])
                tok_c = S.CompactFn.apply(tok, part)
                pos = _pos(self.encoder_pos_embeds[i], cen_c)
                y_c = self.encoder_blocks[i](tok_c, pos.to(tok_c.dtype), bits)
                prev = S.MergeFn.apply(y_c, tok, part)            # a masked token hands on its un-encoded embedding
                outs.append(prev)
                continue
            from . import heads
            with torch.no_grad():       # == pack_mask(~(vis_i & vis_j) | radius_mask(centres)), one launch
                bits = ops.radius_mask_bits(centers[i], None, self.local_radius[i], masked=masks[i])
            pos = _pos(self.encoder_pos_embeds[i], centers[i])
            y = self.encoder_blocks[i](tok, pos.to(tok.dtype), bits)
            outs.append(y)
            prev = heads.where_rows(masks[i], y, tok)              # a masked token hands on its un-encoded embedding
        return outs

    def forward(self, pts, mask=None, group=None, noaug=False, vis_count=None, prop=None):
        """pts (B,N,3) f32; mask (B,64) bool over the COARSEST tokens (True = masked; None: nothing masked).
        vis_count: the number of visible coarsest tokens of EVERY cloud when the caller knows it (generate_mask_ids keeps exactly
        len_keep): a static bound for the visible-first order; a wrong bound sets masked_stack.overflow_flag.
        -> dict: rec (B,256,k1,3) reconstructed level-1 patches, loss_pred (B,64), masks (per level), group, features."""
        neighborhoods, centers, idxs = group if group is not None else self.group_divider(pts)
        B = centers[0].shape[0]
        compact = VISIBLE_FIRST and STACK_NODE and FUSED_BLOCKS and mask is not None and centers[0].is_cuda
        if mask is None:
            mask = torch.zeros(B, self.num_group, dtype=torch.bool, device=centers[0].device)
        masks = back_project(mask, idxs)
        enc = self.encode(neighborhoods, centers, idxs, masks, vis_count=vis_count, compact=compact)
        x2 = self.encoder_norms[2](enc[2])
        if noaug:
            return x2
        from . import heads
        # the mask token's gradient = a column sum over the masked rows: our own kernel (replay-safe), not torch's reduction
        xc = heads.where_rows(masks[2], x2, self.mask_token)
        xc = self.h_decoder[0](xc, _pos(self.decoder_pos_embeds[0], centers[2]).to(xc.dtype))
        h = self.loss_pred_head
        if xc.is_cuda:
            # the north-star model's fused head (heads.LossPredHeadFn): its reductions are our own kernels -- the bias gradient in
            # front of the BatchNorm (a column sum over B*64 rows of 1024) is the one PyTorch reduction of this model that comes
            # back wrong from a hipGraph replay at B = 128 (tools/m2ae_step_diag.py)
            from . import heads
            meta = {"adt": heads._adt(), "training": h[1].training, "eps": h[1].eps, "momentum": h[1].momentum,
                    "slope": h[2].negative_slope, "grad": torch.is_grad_enabled()}
            meta["act_taps"] = ACT_TAPS if torch.is_grad_enabled() else None
            loss_pred = heads.LossPredHeadFn.apply(xc, h[0].weight, h[0].bias, h[1].weight, h[1].bias, h[3].weight, h[3].bias,
                                                   h[1].running_mean, h[1].running_var, h[1].num_batches_tracked, meta)
        else:
            y = F.leaky_relu(h[1](F.linear(xc.reshape(B * self.num_group, -1), h[0].weight.squeeze(-1), h[0].bias)), h[2].negative_slope)
            loss_pred = F.linear(y, h[3].weight.squeeze(-1), h[3].bias).mean(dim=-1).view(B, self.num_group)
        x1 = self.encoder_norms[1](enc[1])
        x1 = heads.where_rows(masks[1], x1, None)
        x1 = self.token_prop[0](centers[1], centers[2], x1, xc, nbrs=prop)
        x1 = self.h_decoder[1](x1, _pos(self.decoder_pos_embeds[1], centers[1]).to(x1.dtype))
        x1 = self.decoder_norm(x1)
        G1, k1 = neighborhoods[1].shape[1], neighborhoods[1].shape[2]
        rec = _linear(x1, self.rec_head.weight, self.rec_head.bias).view(B, G1, k1, 3)
        return {"rec": rec, "loss_pred": loss_pred, "masks": masks, "group": (neighborhoods, centers, idxs), "features": x2}

    def forward_loss(self, rec, neighborhoods, idxs, masks):
        """Chamfer-L2 between the reconstructed and the true level-1 patches (k1 points each), averaged over the MASKED level-1
        tokens; `matrix` (B,64): per coarsest token, the mean patch loss of its masked level-1 members (the loss predictor's
        target)."""
        B, G1, k1, _ = rec.shape
        per_point = self.loss_func(rec.reshape(B * G1, k1, 3).float(), neighborhoods[1].reshape(B * G1, k1, 3).float())   # d1 + d2
        cd = per_point.view(B, G1, k1).mean(dim=-1)                                   # (B,256)
        m1 = masks[1].to(cd.dtype)
        loss = (cd * m1).sum() / m1.sum().clamp_min(1.0)
        member = idxs[2]                                                              # (B,64,k2) level-1 members of a coarse token
        mm = M.take(m1, member.reshape(B, -1)).view(member.shape)
        mc = M.take(cd, member.reshape(B, -1)).view(member.shape)
        matrix = (mc * mm).sum(dim=-1) / mm.sum(dim=-1).clamp_min(1.0)
        return {"Chamfer_mean": loss, "matrix": matrix, "per_token": cd}


def group_stage(divider, pts, augment=True):
    """Everything of an iteration that depends on the clouds alone -- augmentation (P/engine_pretrain.py:80), the three FPS + KNN levels,
    the token propagation's 3-NN lists -- -> (augmented clouds, (neighbourhoods, centres, idxs), (prop_idx, prop_w)).  No parameter is
    read: GraphedM2AEStep replays this for the NEXT batch on a second stream while the current batch trains."""
    from . import engine_pretrain as E
    with torch.no_grad():
        if augment:
            pts = E.train_transforms(pts)
        group = divider(pts)
        prop = TokenPropagation.neighbours(group[1][1], group[1][2])
    return pts, group, prop


def pretrain_forward(model, teacher, pts, epoch, total_epoch, mask_noise=None, group=None, prop=None):
    """One GeoMask3D iteration's forward on the hierarchical model: the EMA teacher scores the 64 coarsest tokens with nothing
    masked, the guided mask (P/models_mae_learn_loss.py:744-784 with this model's mask ratio) hides the hardest ones, the student
    reconstructs and predicts its own per-token loss.  -> dict with `loss`, `loss_chfr`, `loss_learn`, `mask`.
    group / prop: the results of group_stage when the caller has them already."""
    raw = model.module if hasattr(model, "module") else model
    with torch.no_grad():
        group = group if group is not None else teacher.group_divider(pts)
        if prop is None:
            prop = TokenPropagation.neighbours(group[1][1], group[1][2])     # shared by the teacher's and the student's up-block
        t = teacher(pts, mask=None, group=group, prop=prop)
        mask, vis_ids, mask_ids = M.generate_mask_ids(t["loss_pred"], mask_ratio=raw.mask_ratio, guide=True, epoch=epoch,
                                                      total_epoch=total_epoch, noise=mask_noise)
        masked = mask.to(torch.bool)
    out = model(pts, mask=masked, group=group, vis_count=vis_ids.shape[1], prop=prop)
    lo = raw.forward_loss(out["rec"], group[0], group[2], out["masks"])
    pred = M.take(out["loss_pred"].float(), mask_ids)
    target = M.take(lo["matrix"].detach().float(), mask_ids)
    from . import heads
    loss_learn = heads.rank_loss(pred, target)
    res = {"loss": lo["Chamfer_mean"] + loss_learn, "loss_chfr": lo["Chamfer_mean"], "loss_learn": loss_learn, "mask": masked,
           "teacher_loss_pred": t["loss_pred"], "matrix": lo["matrix"], "rec": out["rec"]}
    if pts.is_cuda and VISIBLE_FIRST:
        # (1,) int32 device flag, non-zero if a cloud ever had more visible tokens than the static bound of the visible-first order
        # (tokens were then dropped): read it where the finite-loss guard is read, never inside a step
        from . import masked_stack
        res["vis_overflow"] = masked_stack.overflow_flag(pts.device)
    return res


def pretrain_step(model, model_ema, optimizer, pts, epoch, args, mask_noise=None, augment=True, staged=None):
    """augment -> teacher -> mask -> student -> losses -> backward -> clip(5) -> AdamW -> EMA (the loop body of
    P/engine_pretrain.py:77-212 around this model).  staged = group_stage(...)'s result for these clouds (then `pts`, `augment` are unused)."""
    from . import engine_pretrain as E
    from contextlib import nullcontext
    group = prop = None
    if staged is not None:
        pts, group, prop = staged
    elif augment:
        pts = E.train_transforms(pts)
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if getattr(args, "bf16", False) else nullcontext()
    with amp:
        out = pretrain_forward(model, model_ema.ema, pts, epoch, args.epochs, mask_noise=mask_noise, group=group, prop=prop)
    optimizer.zero_grad(set_to_none=True)
    if DEFER_WGRADS and pts.is_cuda:
        from . import fused
        with fused.async_wgrad(pts.device, defer_heads=True):       # the 19 small layers' weight gradients: one launch at the exit
            out["loss"].backward()
    else:
        out["loss"].backward()
    out["grad_norm"] = E.step_update(model, model_ema, optimizer)
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}


class GraphedM2AEStep:
    """The Point-M2AE + GeoMask3D step as hipGraph replays, with the NEXT batch's grouping beside the current batch's training.

    Two graphs: `group_graph` = group_stage (augmentation, three FPS + KNN levels, 3-NN lists: ~1.3 ms of one-workgroup-per-cloud
    chains that use half the CUs and no parameter) on a second stream; `train_graph` = everything else (teacher, mask, student, losses,
    backward, clip + AdamW + EMA).  A call copies the staged results into the training graph's own inputs (one multi-tensor copy),
    queues the next batch's grouping on the side stream and replays the training graph -- what engine_finetune.GraphedFinetuneStep does
    for its point sampling.  Without `next_pts` the grouping of a batch runs right before its training (same results: the stages read
    and write disjoint state; with augmentation on, the random draws of a batch are taken when its grouping is queued).
    `epoch` is baked into the captures (mask schedule), like GraphedPretrainStep."""

    def __init__(self, model, model_ema, optimizer, args, example, epoch, augment=True, inject_mask_noise=False, warmup_iters=2):
        from . import streams
        self.model, self.ema, self.opt, self.args, self.epoch, self.augment = model, model_ema, optimizer, args, epoch, augment
        raw = model.module if hasattr(model, "module") else model
        dev = example.device
        self.static_in = example.clone()
        self.static_noise = torch.rand(example.shape[0], raw.num_group, device=dev) if inject_mask_noise else None
        self.side = torch.cuda.Stream(device=dev)
        self._ready, self._staged = torch.cuda.Event(), False
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                pretrain_step(model, model_ema, optimizer, self.static_in.clone(), epoch, args, mask_noise=self.static_noise, augment=augment)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        divider = model_ema.ema.group_divider
        self.group_graph = torch.cuda.CUDAGraph()
        with streams.capture(self.group_graph, stream=self.side):
            pts, group, prop = group_stage(divider, self.static_in, augment=augment)
            self._stage_out = [pts] + list(group[0]) + list(group[1]) + list(group[2]) + list(prop)
        torch.cuda.synchronize()
        self.group_graph.replay()            # real values for the capture below (a capture executes nothing)
        torch.cuda.synchronize()
        self._train_in = [t.clone() for t in self._stage_out]
        n = len(group[0])
        ti = self._train_in
        staged = (ti[0], (ti[1:1 + n], ti[1 + n:1 + 2 * n], ti[1 + 2 * n:1 + 3 * n]), (ti[1 + 3 * n], ti[2 + 3 * n]))
        self.train_graph = torch.cuda.CUDAGraph()
        with streams.capture(self.train_graph):
            self.out = pretrain_step(model, model_ema, optimizer, None, epoch, args, mask_noise=self.static_noise, staged=staged)
        torch.cuda.synchronize()

    def _enqueue_grouping(self, pts):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())       # after the copy that consumed the previous staged batch, and after pts' producer
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.static_in.copy_(pts, non_blocking=True)
            pts.record_stream(self.side)
            self.group_graph.replay()
            self._ready.record(self.side)
        self._staged = True

    def __call__(self, pts, mask_noise=None, next_pts=None):
        main = torch.cuda.current_stream()
        if not self._staged:                          # first call, or no look-ahead was given last time: group this batch now
            self._enqueue_grouping(pts)
        main.wait_event(self._ready)
        torch._foreach_copy_(self._train_in, self._stage_out)
        if self.static_noise is not None and mask_noise is not None:
            self.static_noise.copy_(mask_noise, non_blocking=True)
        self._staged = False
        if next_pts is not None:                      # queued BEFORE the training graph so that the two run side by side
            self._enqueue_grouping(next_pts)
        self.train_graph.replay()
        return self.out
