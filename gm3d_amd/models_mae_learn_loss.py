"""Point-MAE + GeoMask3D model on the MI355X-native operators.

Host-side mirror of the reference's Point-MAE_SA3D/models_mae_learn_loss.py (P/ below) for the
pretrain hot path: same class / method names, argument meaning, returned dict keys and
state-dict key names for every LIVE parameter, so engine_pretrain.py-style callers and the
fine-tune checkpoint loader (P/main_finetune.py:311-324) work unchanged.  Differences are of
execution, not of results:

  * FPS, KNN+grouping, Chamfer and the attention core run as hand-written HIP kernels behind the
    C ABI (gm3d_amd/ops.py); nothing here has a CPU fallback.
  * the 53.6 M dead image-MAE parameters of the reference (P/:56-94,144-186) are not
    instantiated -- they never receive a gradient and never influence an output.
  * boolean-mask gathers (P/:298-299,649-650, host sync) are index gathers with a static visible
    count; the shared pos_embed MLP is evaluated once for all 64 centres instead of three times;
    the loss-predictor head's Conv1d(1024,384) + mean(-1) is folded into one 1024-vector product;
    the mini-PointNet's concat([global,local]) @ W is evaluated as local @ W_l + global @ W_g.
    Each of these is an exact algebraic identity (fp32 rounding-order differences only).
"""
from functools import partial

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from . import streams
from .ops import ChamferDistanceL1, ChamferDistanceL2  # noqa: F401  (re-exported like the reference imports)


def drop_path(x, p, training):
    """timm-0.4.5 DropPath: x / keep * floor(keep + U[0,1)), one draw per sample."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    m = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor_()
    return x.div(keep) * m


def drop_path_scale(B, p, training, device):
    """Per-sample DropPath factor floor(keep + U[0,1)) / keep as an f32 (B,) vector, or None when inactive:
    the form the fused transformer stack consumes (gm3d_amd/fused.py)."""
    if p == 0.0 or not training:
        return None
    keep = 1.0 - p
    return (keep + torch.rand(B, dtype=torch.float32, device=device)).floor_().div_(keep)


_default_drop_path_scale = drop_path_scale
_zero_cache = {}


def _zero_scalar(device):
    """a constant f32 0-d zero on `device`, created once OUTSIDE any stream capture (a tensor first filled inside a capture holds
    nothing until the first replay); while capturing without one, a fresh zeros(()) as before."""
    key = str(device)
    z = _zero_cache.get(key)
    if z is None:
        z = torch.zeros((), dtype=torch.float32, device=device)
        if not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _zero_cache[key] = z
    return z


_keep_cache = {}
_dp_pool = []          # [(B, probs tuple, [scale | None per site])] drawn ahead by prepare_drop_path, consumed in order


def _draw_scales(B, plist, device):
    """One uniform draw for every active DropPath site of the stacks in `plist` (lists of per-site probabilities):
    3 launches in all.  -> per stack, a list with an f32 (B,) factor or None per site."""
    flat = [(si, i, p) for si, probs in enumerate(plist) for i, p in enumerate(probs) if p > 0.0]
    outs = [[None] * len(probs) for probs in plist]
    if flat:
        key = (tuple(p for _, _, p in flat), str(device))
        keep = _keep_cache.get(key)
        if keep is None:   # built once (outside any stream capture: the eager warm-up steps come first)
            keep = torch.tensor([1.0 - p for p in key[0]], dtype=torch.float32).unsqueeze(1).to(device)
            _keep_cache[key] = keep
        u = torch.rand(len(flat), B, dtype=torch.float32, device=device)
        if u.is_cuda:         # floor(keep + u) / keep: the same three IEEE operations in one launch
            from ._capi import lib
            s = torch.empty_like(u)
            ops._launch("gm3d_drop_path_scales", {"S": len(flat), "B": B}, lib.gm3d_drop_path_scales, ops._ptr(u), ops._ptr(keep), len(flat), B,
                        ops._ptr(s), ops._stream())
        else:
            s = (u + keep).floor_().div_(keep)
        for j, (si, i, _) in enumerate(flat):
            outs[si][i] = s[j]
    return outs


def prepare_drop_path(B, plist, training, device):
    """Draw the DropPath factors of several stacks that are about to run (the student's encoder and its two decoders) in one
    go; drop_path_scales hands them out in this order.  No-op in eval mode or when drop_path_scale has been replaced."""
    del _dp_pool[:]
    if training and drop_path_scale is _default_drop_path_scale:
        for probs, out in zip(plist, _draw_scales(B, plist, device)):
            _dp_pool.append((B, tuple(probs), out))


def drop_path_scales(B, probs, training, device):
    """DropPath factors for a whole stack (two per block): one uniform draw of shape (n_active, B) instead of one per
    site.  Falls back to per-site calls when drop_path_scale has been replaced (the tests replay recorded draws)."""
    if drop_path_scale is not _default_drop_path_scale:
        del _dp_pool[:]
        return [drop_path_scale(B, p, training, device) for p in probs]
    if not training:
        return [None] * len(probs)
    if _dp_pool:
        pb, pp, out = _dp_pool[0]
        if pb == B and pp == tuple(probs):
            _dp_pool.pop(0)
            return out
        del _dp_pool[:]          # out of step with the plan: drop it
    return _draw_scales(B, [probs], device)[0]


# the student's two decoders are independent given x_full: the loss-prediction decoder runs on a second HIP stream (forward, and
# backward by autograd on the same stream; a captured graph keeps the fork / join as parallel branches).  +3.5 % on the step.
PARALLEL_DECODERS = True      # module attribute (the equality tests flip it), not an environment switch
VISIBLE_EMBED = True          # student: last embed conv on the visible groups only (tests flip it)
_decoder_streams = {}


def _decoder_stream(device):
    key = (device.type, device.index)
    if key not in _decoder_streams:
        _decoder_streams[key] = torch.cuda.Stream(device=device)
    return _decoder_streams[key]


FUSED_STACK = True   # False: per-op PyTorch modules below (kept as the in-package cross-check of the fused path)
FUSED_EMBED = True   # same switch for the mini-PointNet token embed (gm3d_amd/embed.py)
FUSED_HEADS = True   # ... and for pos_embed, the two heads, the mask-token expand and the ranking loss (gm3d_amd/heads.py)


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    """timm Attention (in-tree twin P/models/Point_MAE.py:101-125) with the softmax(QK^T)V core on MFMA."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        if dim // num_heads != 64 or attn_drop != 0.0 or proj_drop != 0.0:
            raise NotImplementedError("the HIP attention core is built for head_dim 64 without dropout")
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        return self.proj(ops.attention(self.qkv(x), self.num_heads, self.scale))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


def linear3(x, weight, bias):
    """y = x @ weight^T + bias for in_features == 3 (xyz inputs), as three broadcast FMAs in fp32.
    hipBLASLt is kept away from K=3 on purpose: a 3-element bf16 row is 6 bytes, no vector load is aligned,
    and under hipGraph replay the K=3 GEMM path was observed to corrupt neighbouring small allocations."""
    x = x.float()
    w = weight.float()
    y = x[..., 0:1] * w[:, 0] + x[..., 1:2] * w[:, 1] + x[..., 2:3] * w[:, 2]
    return y + bias.float()


class Encoder(nn.Module):
    """mini-PointNet token embed (P/:868-899).  Parameters keep the Conv1d/BatchNorm1d layout and names
    (encoder.first_conv.{0,1,3}, encoder.second_conv.{0,1,3}); the math runs on a (rows, channels)
    layout so no transposes are materialised."""

    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, self.encoder_channel, 1))

    def fused(self, point_groups):
        return FUSED_EMBED and point_groups.is_cuda and (self.training or not torch.is_grad_enabled()
                                                         or not any(p.requires_grad for p in self.parameters()))

    def forward(self, point_groups, vis_ids=None):
        """vis_ids (B,V) int64 (optional): return only these groups' tokens, (B,V,C) = take(forward(point_groups), vis_ids); the
        fused path then runs its last conv on those groups alone (embed.EmbedFn)."""
        if self.fused(point_groups):
            from . import embed
            return embed.run_embed(self, point_groups, vis_ids=vis_ids)
        if vis_ids is not None:
            return take(self.forward(point_groups), vis_ids)
        bs, g, n, _ = point_groups.shape
        c0, bn0, _, c1 = self.first_conv
        c2, bn1, _, c3 = self.second_conv
        x = point_groups.reshape(bs * g * n, 3)
        h = F.relu(bn0(linear3(x, c0.weight.squeeze(-1), c0.bias)))
        f = F.linear(h, c1.weight.squeeze(-1), c1.bias)                      # (rows, 256)
        fg = f.view(bs * g, n, 256).amax(dim=1)                              # (groups, 256)
        w2 = c2.weight.squeeze(-1)                                           # [:, :256] global | [:, 256:] local
        y = F.linear(f, w2[:, 256:]).view(bs * g, n, 512) + F.linear(fg, w2[:, :256], c2.bias).unsqueeze(1)
        y = F.relu(bn1(y.view(bs * g * n, 512)))
        z = F.linear(y, c3.weight.squeeze(-1), c3.bias).view(bs * g, n, self.encoder_channel).amax(dim=1)
        return z.view(bs, g, self.encoder_channel)


class TransformerEncoder(nn.Module):
    def __init__(self, embed_dim=768, depth=4, num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0):
        super().__init__()
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate,
                  drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate)
            for i in range(depth)])

    def forward(self, x, pos, norm=None):
        """`norm`: the LayerNorm the caller applies right after (norm_p, P/:303); passing it lets the whole
        stack + norm run as one fused autograd node."""
        if FUSED_STACK and norm is not None and x.is_cuda:
            from . import fused
            return fused.run_stack(self.blocks, norm, x, pos, self.training)
        for block in self.blocks:  # pos is re-added before EVERY block (P/:914-917)
            x = block(x + pos)
        return x if norm is None else norm(x)


class TransformerDecoder(nn.Module):
    def __init__(self, embed_dim=384, depth=4, num_heads=6, mlp_ratio=4.0, qkv_bias=False, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm):
        super().__init__()
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate,
                  drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.apply(self._init_weights)

    def _init_weights(self, m):  # P/:975-982
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x, pos, return_token_num):
        if FUSED_STACK and x.is_cuda:
            from . import fused
            return self.head(fused.run_stack(self.blocks, self.norm, x, pos, self.training))
        for block in self.blocks:
            x = block(x + pos)
        return self.head(self.norm(x))  # ALL tokens, like P/:989


class Group(nn.Module):
    """FPS + KNN grouping (P/:919-957): one FPS launch (centres gathered in-kernel) and one fused
    KNN + gather + centre-subtract launch."""

    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group = num_group
        self.group_size = group_size
        self.knn = ops.KNN(k=self.group_size, transpose_mode=True)

    def fps(self, data, number):
        """data (B,N,3) -> sampled points (B,number,3) (P/:926-933)."""
        return ops.fps(data.contiguous(), number)[1]

    def forward(self, xyz):
        xyz = xyz.contiguous()
        center = self.fps(xyz, self.num_group)
        neighborhood, neighborhood_org, idx = ops.knn_group(xyz, center, self.group_size, return_idx=True)
        assert idx.size(1) == self.num_group and idx.size(2) == self.group_size
        return neighborhood, center, neighborhood_org


def split_ids(mask, num_visible=None):
    """(B,L) bool -> (ids of visible tokens, ids of masked tokens), each in original index order:
    the order boolean-mask indexing produces at P/:298-299,649-650."""
    if num_visible is None:
        num_visible = int((~mask[0]).sum())  # host sync; the engine passes the static count instead
    ids = torch.argsort(mask.to(torch.uint8), dim=1, stable=True)
    return ids[:, :num_visible], ids[:, num_visible:]


def take(x, ids):
    """x (B,L,...) gathered along dim 1 by ids (B,K)."""
    tail = x.shape[2:]
    ix = ids.reshape(ids.shape + (1,) * len(tail)).expand(ids.shape + tail)
    return torch.gather(x, 1, ix)


@torch.no_grad()
def generate_mask_ids(loss_pred, mask_ratio=0.75, guide=True, epoch=0, total_epoch=200, noise=None, want_bool=False):
    """Teacher-guided mask (P/:744-784) plus the id lists split_ids would derive from it, as ONE launch (gm3d_mask_select):
    -> (mask (B,L) f32 0 keep / 1 remove, vis_ids (B,len_keep) int64, mask_ids (B,L-len_keep) int64[, the mask as bool]).  L <= 64."""
    from ._capi import lib
    N, L = loss_pred.shape
    len_keep = int(L * (1 - mask_ratio))
    keep_ratio = float((epoch + 1) / total_epoch) * 0.5 if guide else 0.5
    len_loss = int((L - len_keep) * keep_ratio)
    dev = loss_pred.device
    if noise is None:
        noise = torch.rand(N, L, device=dev)
    noise = noise.to(dev, torch.float32).contiguous()
    lp = loss_pred.detach().float().contiguous()
    mask = torch.empty(N, L, dtype=torch.float32, device=dev)
    order = torch.empty(N, L, dtype=torch.int64, device=dev)        # [visible ids | masked ids] per sample
    vis_ids, mask_ids = order[:, :len_keep], order[:, len_keep:]
    mask_b = torch.empty(N, L, dtype=torch.bool, device=dev) if want_bool else None
    ops._launch("gm3d_mask_select", {"B": N, "L": L}, lib.gm3d_mask_select_b, ops._ptr(lp), ops._ptr(noise), N, L, len_keep,
                len_loss, ops._ptr(mask), ops._ptr(mask_b), ops._ptr(vis_ids), ops._ptr(mask_ids), L, ops._stream())
    if want_bool:
        return mask, vis_ids, mask_ids, mask_b
    return mask, vis_ids, mask_ids


class MaskedAutoencoderViT(nn.Module):
    """GM3D Point-MAE (P/:30-188 live part).  The image-MAE constructor arguments are accepted and
    ignored, exactly as the reference ignores them for the point-cloud path (hyper-parameters are the
    literals of P/:110-117)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0,
                 norm_layer=nn.LayerNorm, norm_pix_loss=False, asymmetric_decoder=False, mask_ratio=0.75,
                 vis_mask_ratio=0.0, saliency=False):
        super().__init__()
        self.norm_pix_loss = norm_pix_loss
        self.vis_mask_ratio = vis_mask_ratio
        self.encoder_dims = 384
        self.trans_dim = 384
        self.depth = 12
        self.drop_path_rate = 0.1
        self.num_heads = 6
        self.group_size = 32
        self.decoder_depth = 4
        self.decoder_num_heads = 6
        self.num_group = 64

        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, 384))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads)
        # both decoders take the FIRST four entries of the 12-long list (P/:119,129,135)
        self.MAE_decoder = TransformerDecoder(embed_dim=self.trans_dim, depth=self.decoder_depth,
                                              drop_path_rate=dpr, num_heads=self.decoder_num_heads)
        self.MAE_decoder_loss_pred = TransformerDecoder(embed_dim=self.trans_dim, depth=self.decoder_depth,
                                                        drop_path_rate=dpr, num_heads=self.decoder_num_heads)
        self.norm_p = nn.LayerNorm(self.trans_dim)
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.increase_dim_2 = nn.Sequential(nn.Conv1d(self.trans_dim, 1024, 1), nn.BatchNorm1d(1024),
                                            nn.LeakyReLU(negative_slope=0.2),
                                            nn.Conv1d(1024, self.trans_dim, 1, bias=True))
        self.increase_dim_just_network_without_feature = nn.Sequential(
            nn.Conv1d(self.trans_dim, 3 * self.group_size, 1, bias=True))
        self.loss_func = ChamferDistanceL2()

    # ------------------------------------------------------------------ forward pieces
    def embed_pos(self, center):
        """self.pos_embed (Linear(3,128) -> GELU -> Linear(128,384), P/:104-108) with the K=3 layer as FMAs."""
        l0, act, l1 = self.pos_embed
        if FUSED_HEADS and center.is_cuda:
            from . import heads
            return heads.PosEmbedFn.apply(center, l0.weight, l0.bias, l1.weight, l1.bias, heads._adt())
        return l1(act(linear3(center, l0.weight, l0.bias)))

    def _encode_visible(self, neighborhood, vis_ids, pos_all, tokens=None):
        if tokens is None:
            tokens = self.encoder(neighborhood)  # B G C
        return self.blocks(take(tokens, vis_ids), take(pos_all, vis_ids), norm=self.norm_p)

    def forward_encoder_point(self, neighborhood, center, mask, num_visible=None):
        """P/:293-306: embed -> keep visible tokens -> pos -> 12 blocks -> norm_p."""
        vis_ids, _ = split_ids(mask, num_visible)
        return self._encode_visible(neighborhood, vis_ids, self.embed_pos(center))

    def _loss_pred_head(self, x):
        """increase_dim_2 then mean over channels (P/:668,677): Conv1d(384,1024) -> BN1d -> LeakyReLU ->
        [Conv1d(1024,384) ; mean(-1)] with the last two folded into one 1024-vector."""
        c0, bn, act, c1 = self.increase_dim_2
        B, L, C = x.shape
        if FUSED_HEADS and x.is_cuda and (self.training or not torch.is_grad_enabled() or not c0.weight.requires_grad):
            from . import heads
            meta = {"adt": heads._adt(), "training": bn.training, "eps": bn.eps, "momentum": bn.momentum,
                    "slope": act.negative_slope, "grad": torch.is_grad_enabled()}
            return heads.LossPredHeadFn.apply(x, c0.weight, c0.bias, bn.weight, bn.bias, c1.weight, c1.bias,
                                              bn.running_mean, bn.running_var, bn.num_batches_tracked, meta)
        h = act(bn(F.linear(x.reshape(B * L, C), c0.weight.squeeze(-1), c0.bias)))
        w = c1.weight.squeeze(-1).mean(dim=0)
        return (F.linear(h, w.unsqueeze(0)).squeeze(-1) + c1.bias.mean()).view(B, L)

    def forward(self, pts, mask, noaug=False, num_visible=None, group=None, need_pix_pred=True, tokens=None,
                pos_all=None, ids=None, cut=False, tokens_visible=False):
        """pts (B,N,3) f32, mask (B,64) bool (True = masked).  Extra keyword-only conveniences for the
        engine: `num_visible` (static visible count, avoids a host sync), `group` (a previously
        computed (neighborhood, center, neighborhood_org), e.g. the teacher's -- the student sees the
        identical samples), `need_pix_pred=False` (skip the reconstruction decoder whose output the
        teacher pass never reads, P/engine_pretrain.py:86-94), `tokens` / `pos_all` (this model's token embed and
        positional embed of all 64 groups when the engine has already evaluated them -- neither depends on the mask;
        tokens_visible=True: `tokens` is (B,V,C), already restricted to the visible groups in vis_ids order)."""
        neighborhood, center, neighborhood_org = group if group is not None else self.group_divider(pts)
        vis_ids, mask_ids = ids if ids is not None else split_ids(mask, num_visible)   # ids: precomputed (generate_mask_ids)
        if pos_all is None:
            pos_all = self.embed_pos(center)
        pos_full = None
        if self.training and FUSED_STACK and pos_all.is_cuda and not noaug:
            # the DropPath factors of the three stacks that follow, from one uniform draw (3 launches instead of 9)
            sites = lambda stack: [p for b in stack.blocks for p in (getattr(b.drop_path, "drop_prob", 0.0),) * 2]
            dec = [sites(self.MAE_decoder)] if need_pix_pred else []
            prepare_drop_path(pts.shape[0], [sites(self.blocks)] + dec + [sites(self.MAE_decoder_loss_pred)], True, pos_all.device)
        if FUSED_HEADS and pos_all.is_cuda:
            # visible-token gather, its positional gather and the [visible | masked] positional concat in one launch
            from . import heads
            if tokens is None and VISIBLE_EMBED and vis_ids.shape[1] < self.num_group and self.encoder.fused(neighborhood):
                # the mask is known: embed the visible groups only (the other 39 tokens of the reference's x = encoder(...) are
                # never read, P/:298) -- same values, 61 % less work in the last conv and its backward
                tokens, tokens_visible = self.encoder(neighborhood, vis_ids=vis_ids), True
            if tokens is None:
                tokens = self.encoder(neighborhood)
            if tokens_visible:
                pos_vis, pos_full = heads.pos_assemble(pos_all, vis_ids, mask_ids)
                tok_vis = tokens
            else:
                tok_vis, pos_vis, pos_full = heads.token_assemble(tokens, pos_all, vis_ids, mask_ids)
            x_vis = self.blocks(tok_vis, pos_vis, norm=self.norm_p)
        elif tokens_visible:
            x_vis = self.blocks(tokens, take(pos_all, vis_ids), norm=self.norm_p)
        else:
            x_vis = self._encode_visible(neighborhood, vis_ids, pos_all, tokens)
        B, _, C = x_vis.shape
        if noaug:
            return x_vis
        cut_pair = None
        if cut:
            # segmented backward (engine_pretrain.SegmentedDDPStep): the decoders see detached leaves, so that the backward of
            # "losses + heads + decoders" is a closed autograd graph ending at (x_vis, pos_full); their gradients are fed into
            # the encoder's graph by the engine.  (Asking autograd for the gradient of the un-detached tensors would make it run
            # the encoder's backward as well: the node that produced pos_full also consumes the encoder's input gradients.)
            cut_pair = (x_vis, pos_full)
            x_vis = x_vis.detach().requires_grad_(True)
            pos_full = pos_full.detach().requires_grad_(True) if pos_full is not None else None
        N = mask_ids.shape[1]
        if N == 0 and not torch.is_grad_enabled():
            x_full = x_vis                      # the teacher's all-visible pass: nothing to append (no cast, no concat launch)
        else:
            if FUSED_HEADS and x_vis.is_cuda:
                from . import heads
                mask_tokens = heads.ExpandRowsFn.apply(self.mask_token, B, N, x_vis.dtype)
            else:
                mask_tokens = self.mask_token.expand(B, N, -1).to(x_vis.dtype)
            x_full = torch.cat([x_vis, mask_tokens], dim=1)
        if pos_full is None:
            pos_full = torch.cat([take(pos_all, vis_ids), take(pos_all, mask_ids)], dim=1)

        # The two decoders are independent given x_full: each of their GEMMs (8192 rows) is at most one wave of tiles on
        # 256 CUs, so the loss-prediction decoder runs on a second HIP stream beside the reconstruction decoder (forward
        # here, backward by autograd on the same streams; a captured graph keeps the fork/join as parallel branches).
        side = None
        if PARALLEL_DECODERS and need_pix_pred and x_full.is_cuda:
            main = torch.cuda.current_stream()
            side = _decoder_stream(x_full.device)
            streams.fork(side, main, who="models_mae_learn_loss: loss-prediction decoder branch (PARALLEL_DECODERS)")
            with torch.cuda.stream(side):     # the loss-prediction branch: decoder AND its head
                loss_pred_ = self.MAE_decoder_loss_pred(x_full, pos_full, N)
                loss_pred_out = self._loss_pred_head(loss_pred_)
            x_full.record_stream(side)
            pos_full.record_stream(side)
        rebuild_points = None
        if need_pix_pred:
            x_rec = self.MAE_decoder(x_full, pos_full, N)
            c = self.increase_dim_just_network_without_feature[0]
            if FUSED_HEADS and x_rec.is_cuda:
                from . import heads
                rebuild_points = heads.LinearBiasFn.apply(x_rec, c.weight, c.bias, heads._adt())
            else:
                rebuild_points = F.linear(x_rec, c.weight.squeeze(-1), c.bias)  # B L 96
        if side is not None:
            streams.join(side)
            loss_pred_out.record_stream(torch.cuda.current_stream())
        else:
            loss_pred_out = self._loss_pred_head(self.MAE_decoder_loss_pred(x_full, pos_full, N))
        return {
            "pix_pred": rebuild_points,
            "mask": mask,
            "mask_num": N,
            "pos_full": pos_full,       # not in the reference dict: the segmented data-parallel backward cuts the graph here
            "cut": cut_pair,            # (x_vis, pos_full) before the cut when cut=True ("features" / "pos_full" are the leaves)
            "features": x_vis,
            "loss_pred": loss_pred_out,
            "neighborhood": neighborhood,
            "neighborhood_org": neighborhood_org,
            "center": center,
        }

    def forward_loss(self, pred, target, mask, mask_ids=None, full_pred=None):
        """pred (B,M,96) = pix_pred[:, -M:], target = neighborhood (B,64,32,3), mask (B,64) bool (P/:384-412).
        full_pred: the whole pix_pred (B,L,96) `pred` was sliced from (optional): the fused loss then takes the slice itself and its
        backward returns the whole tensor's gradient (no separate zero-fill + copy for the slice)."""
        N, t, n, D = target.shape
        M = pred.shape[1]
        if mask_ids is None:
            _, mask_ids = split_ids(mask, t - M)
        if FUSED_HEADS:
            from . import heads
            if heads.patch_chamfer_loss_supported(pred, target, mask_ids):       # gather + cast + Chamfer + both means: one pass
                if full_pred is not None and full_pred.is_contiguous() and full_pred.shape[1] >= M:
                    mean, matrix = heads.PatchChamferLossFn.apply(full_pred, target, mask_ids, True)
                else:
                    mean, matrix = heads.PatchChamferLossFn.apply(pred, target, mask_ids)
                # "MSE_mean" is identically zero in this variant: a cached constant, no launch (the engines read "MSE_zero")
                return {"MSE_mean": _zero_scalar(mean.device), "Chamfer_mean": mean, "matrix": matrix, "MSE_zero": True}
        target = take(target, mask_ids).reshape(-1, n, D).to(torch.float32)
        pred = pred.reshape(-1, n, D).to(torch.float32)
        loss = self.loss_func(pred, target).reshape(N, -1, n)
        matrix = loss.mean(dim=-1)
        mean = matrix.mean()   # == loss.mean() (equal-size groups); keeps the big reduction out of reduce_kernel's multi-block path
        # "MSE_zero": this variant's MSE term is identically zero (P/:384-412) -- the engine skips the dead 13.889 * 0 arithmetic
        return {"MSE_mean": mean.detach() * 0.0, "Chamfer_mean": mean, "matrix": matrix, "MSE_zero": True}

    @torch.no_grad()
    def generate_mask(self, loss_pred, mask_ratio=0.75, images=None, guide=True, epoch=0, total_epoch=200,
                      noise=None):
        """Teacher-guided mask (P/:744-784), vectorised on device, no host sync.
        The `len_loss` tokens with the highest predicted loss are always masked; the other
        L-len_loss tokens are ranked by `noise` (default: fresh U[0,1) draws, which is the uniform
        random permutation np.random.shuffle produces at P/:773) and the first len_keep stay visible.
        Returns float (B,L), 0 = keep, 1 = remove, like the reference."""
        N, L = loss_pred.shape
        len_keep = int(L * (1 - mask_ratio))
        keep_ratio = float((epoch + 1) / total_epoch) * 0.5 if guide else 0.5
        len_loss = int((L - len_keep) * keep_ratio)
        if noise is None:
            noise = torch.rand(N, L, device=loss_pred.device)
        else:
            noise = noise.to(loss_pred.device, torch.float32).clone()
        if loss_pred.is_cuda and L <= 64:
            return self.generate_mask_ids(loss_pred, mask_ratio, guide, epoch, total_epoch, noise)[0]
        if len_loss > 0:
            forced = torch.argsort(loss_pred.float(), dim=1)[:, L - len_loss:]
            noise.scatter_(1, forced, float("inf"))
        keep = torch.argsort(noise, dim=1)[:, :len_keep]
        mask = torch.ones(N, L, device=loss_pred.device)
        mask.scatter_(1, keep, 0.0)
        return mask

    @torch.no_grad()
    def generate_mask_ids(self, loss_pred, mask_ratio=0.75, guide=True, epoch=0, total_epoch=200, noise=None, want_bool=False):
        """generate_mask plus the id lists split_ids would derive from it, as ONE launch (module-level generate_mask_ids)."""
        return generate_mask_ids(loss_pred, mask_ratio, guide, epoch, total_epoch, noise, want_bool)

    def forward_learning_loss(self, loss_pred, mask, loss_target, relative=False, full_pred=None):
        """P/:786-815.  relative=True: pairwise ranking BCE over masked tokens.  full_pred: the (B,L) prediction loss_pred =
        full_pred[:, -M:] was sliced from (optional; the fused loss then folds the slice and its backward in)."""
        loss_pred = loss_pred.float()
        loss_target = loss_target.float()
        if relative and FUSED_HEADS and loss_pred.is_cuda and loss_pred.shape[1] <= 64:
            from . import heads
            if full_pred is not None and full_pred.dtype == torch.float32:
                return heads.rank_loss_tail(full_pred, loss_pred.shape[1], loss_target)
            return heads.rank_loss(loss_pred, loss_target)
        if relative:
            pos = loss_target.unsqueeze(1) > loss_target.unsqueeze(2)
            neg = loss_target.unsqueeze(1) < loss_target.unsqueeze(2)
            sig = torch.sigmoid(loss_pred.unsqueeze(1) - loss_pred.unsqueeze(2))
            loss = -pos.to(sig.dtype) * torch.log(sig + 1e-6) - neg.to(sig.dtype) * torch.log(1 - sig + 1e-6)
            return loss.sum() / (pos | neg).sum()
        mean = loss_target.mean(dim=1, keepdim=True)
        var = loss_target.var(dim=1, keepdim=True)
        return ((loss_pred - (loss_target - mean) / (var + 1.0e-6) ** 0.5) ** 2).mean()

    # ------------------------------------------------------------------ checkpoints
    def load_reference_state_dict(self, state_dict):
        """Load a reference checkpoint's `state_dict`/`model` entry: live keys must all be present and
        match; the reference's dead image-MAE keys (patch_embed.*, decoder_*, increase_dim.*, ...) are
        ignored.  Returns the list of ignored keys."""
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in state_dict.items()}
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError("reference checkpoint lacks live keys: %s" % missing[:5])
        self.load_state_dict({k: sd[k] for k in own}, strict=True)
        return sorted(k for k in sd if k not in own)


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=1024, depth=24, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=14, embed_dim=1280, depth=32, num_heads=16, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


Point_MAE = MaskedAutoencoderViT  # north-star surface name
