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


# ======================================================================================================================
# Point-M2AE + GeoMask3D model, restated functionally over a state dict (plain PyTorch on the CPU + the oracle's FPS / KNN /
# Chamfer).  TEST INFRASTRUCTURE ONLY; "parity unpinned": the reference has no source for this model (M2/README.md:1), this
# follows cfgs/config_Point_M2AE.yaml:57-99, the published Point-M2AE design and the choices listed in
# gm3d_amd/point_m2ae.py's header.  Parameter names are those of gm3d_amd.point_m2ae.PointM2AE's state dict.
import torch.nn.functional as F

CFG = dict(mask_ratio=0.8, group_sizes=(16, 8, 8), num_groups=(512, 256, 64), encoder_depths=(5, 5, 5), encoder_dims=(96, 192, 384),
           local_radius=(0.32, 0.64, 1.28), decoder_depths=(1, 1), decoder_dims=(384, 192), num_heads=6)


def _conv(sd, name, x):
    return x @ sd[name + ".weight"].squeeze(-1).t() + sd[name + ".bias"]


def _bn(sd, name, x, training):
    return F.batch_norm(x, None if training else sd[name + ".running_mean"], None if training else sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=training, eps=1e-5)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


# How far an INJECTED decision is from the one this restatement would have taken itself: every entry is
# (kind, margin relative to the tensor's scale).  A sign taken over against a pre-activation that is not within rounding of zero, or a
# pool winner that is not within rounding of the maximum, would mean the product decided WRONGLY and the comparison merely repeated
# its mistake; tests assert the margins stay at rounding level (tests/test_gpu_m2ae.py).  Reset by the caller.
DECISION_MARGINS = []


class _Signs:
    """The sign patterns of the activations of another run, consumed in call order (None: decide here)."""

    def __init__(self, signs):
        self.signs, self.at = signs, 0

    def act(self, x, slope=0.0):
        if self.signs is None:
            return torch.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)
        s = self.signs[self.at].reshape(x.shape)
        self.at += 1
        with torch.no_grad():
            wrong = s != (x > 0)
            m = float(x.detach().abs()[wrong].max()) / max(float(x.detach().abs().max()), 1e-30) if bool(wrong.any()) else 0.0
            DECISION_MARGINS.append(("sign", m))
        return torch.where(s, x, x * slope)


def _pool(t, win):
    """max over dim 1 of (groups, k, C); with win (groups, k, C) -- 1 at the winner, 1/n at each of n tied winners -- the winner
    is TAKEN from win instead of decided here (a test injects the decisions of the run it compares with: an argmax is a
    discontinuity of the gradient, and exact ties do occur: torch.amax shares the gradient between them)."""
    if win is None:
        return t.max(dim=1)[0]
    out = (t * win.to(t.dtype)).sum(dim=1)
    with torch.no_grad():
        gap = (t.detach().max(dim=1)[0] - out.detach()).abs().max()
        DECISION_MARGINS.append(("pool", float(gap) / max(float(t.detach().abs().max()), 1e-30)))
    return out


def _embed(sd, p, groups, training, pool_idx=None, signs=None):
    signs = signs or _Signs(None)
    B, G, k, C = groups.shape
    x = groups.reshape(-1, C)
    f = _conv(sd, p + ".first_conv.3", signs.act(_bn(sd, p + ".first_conv.1", _conv(sd, p + ".first_conv.0", x), training)))
    f = f.view(B * G, k, -1)
    i0, i1 = (None, None) if pool_idx is None else pool_idx
    y = torch.cat([_pool(f, i0).unsqueeze(1).expand(-1, k, -1), f], dim=2).reshape(B * G * k, -1)
    y = _conv(sd, p + ".second_conv.3", signs.act(_bn(sd, p + ".second_conv.1", _conv(sd, p + ".second_conv.0", y), training)))
    return _pool(y.view(B * G, k, -1), i1).view(B, G, -1)


def _attention(sd, p, x, blocked, H):
    B, T, C = x.shape
    hd = C // H
    qkv = (x @ sd[p + ".qkv.weight"].t()).reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    s = (qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5
    if blocked is not None:
        s = s.masked_fill(blocked.unsqueeze(1), float("-inf"))
    a = torch.nan_to_num(s.softmax(dim=-1), nan=0.0)
    return (a @ qkv[2]).transpose(1, 2).reshape(B, T, C) @ sd[p + ".proj.weight"].t() + sd[p + ".proj.bias"]


def _stack(sd, p, depth, x, pos, blocked, H):
    for i in range(depth):
        q = "%s.blocks.%d" % (p, i)
        x = x + pos
        x = x + _attention(sd, q + ".attn", _ln(sd, q + ".norm1", x), blocked, H)
        h = F.gelu(_ln(sd, q + ".norm2", x) @ sd[q + ".mlp.fc1.weight"].t() + sd[q + ".mlp.fc1.bias"])
        x = x + h @ sd[q + ".mlp.fc2.weight"].t() + sd[q + ".mlp.fc2.bias"]
    return x


def _posmlp(sd, p, c):
    return F.gelu(c @ sd[p + ".0.weight"].t() + sd[p + ".0.bias"]) @ sd[p + ".2.weight"].t() + sd[p + ".2.bias"]


def far_mask(center, radius):
    d2 = torch.zeros(center.shape[0], center.shape[1], center.shape[1])
    for ax in range(3):                                    # coordinate by coordinate, every product and sum rounded to fp32
        diff = center[:, :, None, ax] - center[:, None, :, ax]
        d2 = diff * diff if ax == 0 else d2 + diff * diff
    r = torch.tensor(radius, dtype=torch.float32)
    return d2 >= (r * r)


def m2ae_forward(sd, pts, mask_coarse, training=True, cfg=CFG, group=None, taps=None, pool_idx=None, act_signs=None):
    """sd: the state dict, fp32 -- or fp64 for a high-precision run of the same computation (grouping, masks and the 3-NN search
    stay fp32: they are index decisions the product makes in fp32 as well).  pool_idx: six (groups, k, C) weight tensors, the winners
    of the two max-pools of each level's token embed in call order; act_signs: the nine ReLU / LeakyReLU sign patterns in call
    order (decisions injected by a test; None: decided here)."""
    signs = _Signs(act_signs)
    H = cfg["num_heads"]
    dt = sd["mask_token"].dtype
    nbs, centers, idxs = group if group is not None else hierarchical_group(pts, cfg["num_groups"], cfg["group_sizes"])
    far = [far_mask(centers[i], cfg["local_radius"][i]) for i in range(3)]
    dist, nn_idx = ops.knn(centers[2], centers[1], 3)
    nbs, centers, dist = [t.to(dt) for t in nbs], [t.to(dt) for t in centers], dist.to(dt)
    B = pts.shape[0]
    if mask_coarse is None:
        mask_coarse = torch.zeros(B, cfg["num_groups"][-1], dtype=torch.bool)
    masks = propagate_visibility(mask_coarse, idxs)
    enc, prev = [], None
    for i in range(3):
        if i == 0:
            tok = _embed(sd, "token_embed.0", nbs[0], training, None if pool_idx is None else pool_idx[0:2], signs)
        else:
            feats = torch.stack([prev[b][idxs[i][b]] for b in range(B)])          # (B,G,k,C)
            tok = _embed(sd, "token_embed.%d" % i, feats, training, None if pool_idx is None else pool_idx[2 * i:2 * i + 2], signs)
        if taps is not None and tok.requires_grad:       # diagnostics: the level's embedded tokens, gradient retained
            tok.retain_grad()
            taps["tok%d" % i] = tok
        vis = ~masks[i]
        blocked = ~(vis[:, :, None] & vis[:, None, :]) | far[i]
        y = _stack(sd, "encoder_blocks.%d" % i, cfg["encoder_depths"][i], tok, _posmlp(sd, "encoder_pos_embeds.%d" % i, centers[i]),
                   blocked, H)
        enc.append(y)
        prev = torch.where(vis[..., None], y, tok)
    vis2, vis1 = ~masks[2], ~masks[1]
    x2 = _ln(sd, "encoder_norms.2", enc[2])
    xc = torch.where(vis2[..., None], x2, sd["mask_token"].expand(B, x2.shape[1], -1))
    xc = _stack(sd, "h_decoder.0", cfg["decoder_depths"][0], xc, _posmlp(sd, "decoder_pos_embeds.0", centers[2]), None, H)
    h = xc.reshape(-1, xc.shape[-1])
    h = signs.act(_bn(sd, "loss_pred_head.1", _conv(sd, "loss_pred_head.0", h), training), 0.2)
    loss_pred = _conv(sd, "loss_pred_head.3", h).mean(dim=-1).view(B, -1)
    x1 = torch.where(vis1[..., None], _ln(sd, "encoder_norms.1", enc[1]), torch.zeros((), dtype=dt))
    # token propagation: 3 nearest coarse centres, inverse squared distance weights
    w = 1.0 / (dist * dist + 1e-8)
    w = w / w.sum(dim=-1, keepdim=True)
    near = torch.stack([xc[b][nn_idx[b]] for b in range(B)])                       # (B,256,3,384)
    y = torch.cat([x1, (near * w[..., None]).sum(dim=2)], dim=-1).reshape(B * x1.shape[1], -1)
    for j in range(2):
        y = signs.act(_bn(sd, "token_prop.0.mlp_bns.%d" % j, _conv(sd, "token_prop.0.mlp_convs.%d" % j, y), training))
    x1 = _stack(sd, "h_decoder.1", cfg["decoder_depths"][1], y.view(B, x1.shape[1], -1), _posmlp(sd, "decoder_pos_embeds.1", centers[1]),
                None, H)
    x1 = _ln(sd, "decoder_norm", x1)
    rec = _conv(sd, "rec_head", x1).view(B, x1.shape[1], cfg["group_sizes"][1], 3)
    return {"rec": rec, "loss_pred": loss_pred, "masks": masks, "group": (nbs, centers, idxs), "features": x2}


def m2ae_losses(rec, nbs, idxs, masks, nn_idx=None):
    """nn_idx = (idx1 (P,k1), idx2 (P,k1)) int64: the nearest-neighbour choices of both Chamfer directions, injected by a test (the
    fp64 run takes the fp32 run's: an argmin is a discontinuity of the gradient); None: decided here."""
    B, G1, k1, _ = rec.shape
    a, b = rec.reshape(B * G1, k1, 3), nbs[1].reshape(B * G1, k1, 3).to(rec.dtype)
    if nn_idx is not None:
        pair = (a[:, :, None, :] - b[:, None, :, :]).pow(2).sum(-1)
        d = torch.gather(pair, 2, nn_idx[0].unsqueeze(2)).squeeze(2) + torch.gather(pair, 1, nn_idx[1].unsqueeze(1)).squeeze(1)
    elif rec.dtype == torch.float32:
        d = ops.ChamferDistanceL2()(a, b)                                                       # per point: d1 + d2
    else:                                                                                       # the same quantity in fp64
        pair = (a[:, :, None, :] - b[:, None, :, :]).pow(2).sum(-1)
        d = pair.min(dim=2)[0] + pair.min(dim=1)[0]
    cd = d.view(B, G1, k1).mean(dim=-1)
    m1 = masks[1].to(rec.dtype)
    loss = (cd * m1).sum() / m1.sum().clamp_min(1.0)
    matrix = torch.zeros(B, idxs[2].shape[1], dtype=rec.dtype)
    for b in range(B):
        mm, mc = m1[b][idxs[2][b]], cd[b][idxs[2][b]]                                          # (64,8)
        matrix[b] = (mc * mm).sum(dim=-1) / mm.sum(dim=-1).clamp_min(1.0)
    return loss, matrix, cd


def guided_mask(loss_pred, noise, mask_ratio, epoch, total_epoch):
    """P/models_mae_learn_loss.py:744-784 in its vectorised reading: the len_loss tokens with the highest predicted loss are always
    masked, the rest ranked by `noise`, the first len_keep of that ranking stay visible.  -> bool (B,L), True = masked."""
    B, L = loss_pred.shape
    len_keep = int(L * (1 - mask_ratio))
    len_loss = int((L - len_keep) * (float((epoch + 1) / total_epoch) * 0.5))
    mask = torch.ones(B, L, dtype=torch.bool)
    for b in range(B):
        forced = set(torch.argsort(loss_pred[b])[L - len_loss:].tolist()) if len_loss > 0 else set()
        free = sorted((i for i in range(L) if i not in forced), key=lambda i: float(noise[b, i]))
        mask[b, free[:len_keep]] = False
    return mask


def ranking_loss(pred, target):
    """P/models_mae_learn_loss.py:786-805 (relative=True)."""
    pos = target.unsqueeze(1) > target.unsqueeze(2)
    neg = target.unsqueeze(1) < target.unsqueeze(2)
    d = pred.unsqueeze(1) - pred.unsqueeze(2)
    loss = -pos.to(d.dtype) * torch.log(torch.sigmoid(d) + 1e-6) - neg.to(d.dtype) * torch.log(1 - torch.sigmoid(d) + 1e-6)
    return loss.sum() / (pos | neg).sum()


def m2ae_pretrain_forward(sd, sd_teacher, pts, epoch, total_epoch, noise, cfg=CFG, mask=None, taps=None, decisions=None):
    """mask: use this coarse mask instead of deriving it from the teacher (the fp64 run of a test takes the fp32 run's mask).
    decisions (optional dict): the DISCRETE choices of another run of the same computation, taken over instead of made here, so
    that two runs in different arithmetic differentiate the same piecewise-smooth branch --
      "pool_idx"    the six max-pool winners of the token embeds (m2ae_forward),
      "act_signs"   the nine ReLU / LeakyReLU sign patterns (m2ae_forward),
      "nn_idx"      the Chamfer nearest-neighbour choices (m2ae_losses),
      "rank_target" (B,64): the per-token target whose ORDER decides the sign pattern of the ranking loss (ranking_loss)."""
    decisions = decisions or {}
    with torch.no_grad():
        group = hierarchical_group(pts, cfg["num_groups"], cfg["group_sizes"])
        t = m2ae_forward(sd_teacher, pts, None, training=False, cfg=cfg, group=group)
        if mask is None:
            mask = guided_mask(t["loss_pred"].float(), noise, cfg["mask_ratio"], epoch, total_epoch)
    out = m2ae_forward(sd, pts, mask, training=True, cfg=cfg, group=group, taps=taps, pool_idx=decisions.get("pool_idx"),
                       act_signs=decisions.get("act_signs"))
    loss_cd, matrix, cd = m2ae_losses(out["rec"], group[0], group[2], out["masks"], nn_idx=decisions.get("nn_idx"))
    n_mask = int(mask[0].sum())
    ids = torch.stack([torch.nonzero(mask[b]).flatten() for b in range(pts.shape[0])])              # (B,n_mask) ascending
    assert ids.shape[1] == n_mask
    rank_target = decisions["rank_target"].to(matrix.dtype) if "rank_target" in decisions else matrix.detach()
    loss_learn = ranking_loss(torch.gather(out["loss_pred"], 1, ids), torch.gather(rank_target, 1, ids))
    return {"loss": loss_cd + loss_learn, "loss_chfr": loss_cd, "loss_learn": loss_learn, "mask": mask, "rec": out["rec"],
            "loss_pred": out["loss_pred"], "teacher_loss_pred": t["loss_pred"], "matrix": matrix, "features": out["features"]}
