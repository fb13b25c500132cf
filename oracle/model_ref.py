"""CPU oracle of the GM3D pretrain step's Python glue (plain PyTorch fp32 + oracle/ops.py).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by gm3d_amd/.  Each piece cites the reference lines it restates
(P/ = /root/reference/Point-MAE_SA3D/).  Pinned against the reference's own
models_mae_learn_loss.py by tests/golden/make_golden.py (this container only; the
reference never travels) -> tests/golden/*.npz, checked in tests/test_oracle_golden.py.
The three native ops underneath (FPS, KNN, Chamfer) remain "parity unpinned"
(oracle/gm3d_oracle.c header).

Only the LIVE part of the reference model is restated: the image-MAE leftovers and unused
heads (53.6 M of 90.5 M parameters) never touch the step's results.
"""
import copy
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

# ----------------------------------------------------------------------------- timm 0.4.5 pieces
_droppath_log = None    # when a list: every DropPath keep-mask drawn is appended (fixture capture)
_droppath_feed = None   # when a list: masks are popped from it instead of being drawn


def drop_path(x, p, training):
    """timm 0.4.5 drop_path: x/keep * floor(keep + U[0,1)), one draw per sample."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    if _droppath_feed is not None:
        m = _droppath_feed.pop(0).to(x.dtype).reshape(shape)
    else:
        m = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
    if _droppath_log is not None:
        _droppath_log.append(m.reshape(-1).clone())
    return x.div(keep) * m


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class Mlp(nn.Module):  # P/models/Point_MAE.py:82-98
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):  # P/models/Point_MAE.py:101-125
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class Block(nn.Module):  # P/models/Point_MAE.py:128-146 (timm Block; LayerNorm eps 1e-5)
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


# ----------------------------------------------------------------------------- model
class Encoder(nn.Module):  # P/models_mae_learn_loss.py:868-899
    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, encoder_channel, 1))

    def forward(self, point_groups):
        bs, g, n, _ = point_groups.shape
        f = self.first_conv(point_groups.reshape(bs * g, n, 3).transpose(2, 1))
        fg = torch.max(f, dim=2, keepdim=True)[0]
        f = self.second_conv(torch.cat([fg.expand(-1, -1, n), f], dim=1))
        return torch.max(f, dim=2, keepdim=False)[0].reshape(bs, g, self.encoder_channel)


class TransformerEncoder(nn.Module):  # :901-917  (pos re-added before EVERY block)
    def __init__(self, embed_dim, depth, num_heads, drop_path_rate):
        super().__init__()
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, 4.0, qkv_bias=False, drop_path=drop_path_rate[i])
                                     for i in range(depth)])

    def forward(self, x, pos):
        for blk in self.blocks:
            x = blk(x + pos)
        return x


class TransformerDecoder(nn.Module):  # :959-990 (returns ALL tokens; xavier re-init :973-982)
    def __init__(self, embed_dim, depth, num_heads, drop_path_rate):
        super().__init__()
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, 4.0, qkv_bias=False, drop_path=drop_path_rate[i])
                                     for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x, pos, return_token_num):
        for blk in self.blocks:
            x = blk(x + pos)
        return self.norm(x)


class Group(nn.Module):  # :919-957
    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group, self.group_size = num_group, group_size
        self.knn = ops.KNN(k=group_size, transpose_mode=True)

    def fps(self, data, number):
        idx = ops.furthest_point_sample(data, number)
        return ops.gather_operation(data.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()

    def forward(self, xyz):
        B, N, _ = xyz.shape
        center = self.fps(xyz, self.num_group)
        _, idx = self.knn(xyz, center)
        idx = (idx + torch.arange(0, B).view(-1, 1, 1) * N).view(-1)
        nb_org = xyz.reshape(B * N, -1)[idx, :].view(B, self.num_group, self.group_size, 3).contiguous()
        return nb_org - center.unsqueeze(2), center, nb_org


class PointMAEGM3D(nn.Module):
    """Live part of P/models_mae_learn_loss.py::MaskedAutoencoderViT (:101-188, hyper-parameters :110-117)."""

    def __init__(self, trans_dim=384, depth=12, drop_path_rate=0.1, num_heads=6, group_size=32, num_group=64,
                 decoder_depth=4, decoder_num_heads=6):
        super().__init__()
        self.trans_dim, self.num_group, self.group_size = trans_dim, num_group, group_size
        self.encoder = Encoder(trans_dim)
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, trans_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]  # :119 -- decoders use dpr[0:4]
        self.blocks = TransformerEncoder(trans_dim, depth, num_heads, dpr)
        self.MAE_decoder = TransformerDecoder(trans_dim, decoder_depth, decoder_num_heads, dpr)
        self.MAE_decoder_loss_pred = TransformerDecoder(trans_dim, decoder_depth, decoder_num_heads, dpr)
        self.norm_p = nn.LayerNorm(trans_dim)
        self.group_divider = Group(num_group, group_size)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, trans_dim))  # never re-initialised (:99,141)
        self.increase_dim_2 = nn.Sequential(nn.Conv1d(trans_dim, 1024, 1), nn.BatchNorm1d(1024),
                                            nn.LeakyReLU(negative_slope=0.2), nn.Conv1d(1024, trans_dim, 1))
        self.increase_dim_just_network_without_feature = nn.Sequential(nn.Conv1d(trans_dim, 3 * group_size, 1))
        self.loss_func = ops.ChamferDistanceL2()

    def forward_encoder_point(self, neighborhood, center, mask):  # :293-306
        tokens = self.encoder(neighborhood)
        B, _, C = tokens.shape
        x_vis = tokens[~mask].reshape(B, -1, C)
        pos = self.pos_embed(center[~mask].reshape(B, -1, 3))
        return self.norm_p(self.blocks(x_vis, pos))

    def forward(self, pts, mask, noaug=False):  # :635-684
        neighborhood, center, neighborhood_org = self.group_divider(pts)
        x_vis = self.forward_encoder_point(neighborhood, center, mask)
        B, _, C = x_vis.shape
        if noaug:
            return x_vis
        pos_vis = self.pos_embed(center[~mask]).reshape(B, -1, C)
        pos_mask = self.pos_embed(center[mask]).reshape(B, -1, C)
        N = pos_mask.shape[1]
        x_full = torch.cat([x_vis, self.mask_token.expand(B, N, -1)], dim=1)
        pos_full = torch.cat([pos_vis, pos_mask], dim=1)
        loss_pred = x_full.clone()
        x_rec = self.MAE_decoder(x_full, pos_full, N)
        rebuild = self.increase_dim_just_network_without_feature(x_rec.transpose(1, 2)).transpose(1, 2)
        lp = self.MAE_decoder_loss_pred(loss_pred, pos_full, N)
        lp = self.increase_dim_2(lp.transpose(1, 2)).transpose(1, 2)
        return {"pix_pred": rebuild, "mask": mask, "mask_num": N, "features": x_vis, "loss_pred": lp.mean(dim=-1),
                "neighborhood": neighborhood, "neighborhood_org": neighborhood_org, "center": center}

    def forward_loss(self, pred, target, mask):  # :384-412
        N, _, n, D = target.shape
        target = target[mask].reshape(-1, n, D).float()
        pred = pred.reshape(-1, n, D).float()
        loss = self.loss_func(pred, target).reshape(N, -1, n)
        return {"MSE_mean": loss.mean() * 0.0, "Chamfer_mean": loss.mean(), "matrix": loss.mean(dim=-1)}

    @torch.no_grad()
    def generate_mask(self, loss_pred, mask_ratio=0.75, images=None, guide=True, epoch=0, total_epoch=200,
                      rng=None, noise=None):  # :744-784
        """rng: np.random.RandomState standing in for the global np.random of :773;
        noise: the torch.randn(N, L) of :763 when injected."""
        N, L = loss_pred.shape
        len_keep = int(L * (1 - mask_ratio))
        ids_loss = torch.argsort(loss_pred, dim=1)
        keep_ratio = float((epoch + 1) / total_epoch) * 0.5 if guide else 0.5
        len_loss = int((L - len_keep) * keep_ratio)
        if len_loss <= 0:
            noise = torch.randn(N, L) if noise is None else noise
            ids_shuffle = torch.argsort(noise, dim=1)
        else:
            rng = rng if rng is not None else np.random
            ids_shuffle = torch.zeros_like(ids_loss)
            for i in range(N):
                ids_shuffle[i, -len_loss:] = ids_loss[i, -len_loss:]
                rest = np.delete(np.arange(L), ids_shuffle[i, -len_loss:].numpy())
                rng.shuffle(rest)
                ids_shuffle[i, :L - len_loss] = torch.from_numpy(rest)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        mask = torch.ones(N, L)
        mask[:, :len_keep] = 0
        return torch.gather(mask, dim=1, index=ids_restore)

    def forward_learning_loss(self, loss_pred, mask, loss_target, relative=False):  # :786-815
        if relative:
            pos = loss_target.unsqueeze(1) > loss_target.unsqueeze(2)
            neg = loss_target.unsqueeze(1) < loss_target.unsqueeze(2)
            d = loss_pred.unsqueeze(1) - loss_pred.unsqueeze(2)
            loss = -pos.int() * torch.log(torch.sigmoid(d) + 1e-6) - neg.int() * torch.log(1 - torch.sigmoid(d) + 1e-6)
            return loss.sum() / (pos + neg).sum()
        mean = loss_target.mean(dim=1, keepdim=True)
        var = loss_target.var(dim=1, keepdim=True)
        return ((loss_pred - (loss_target - mean) / (var + 1.0e-6) ** 0.5) ** 2).mean()


# ----------------------------------------------------------------------------- deterministic weights
def det_fill_(model, seed=0):
    """Overwrite every parameter/buffer with values that depend only on (seed, name, shape), so a
    test on another machine can rebuild the exact weights a fixture was made with, without
    shipping 147 MB of them.  Weights ~ U(-a,a), a = sqrt(3/fan_in) (keeps activations O(1));
    norm scales ~ U(0.8,1.2); biases / mask token ~ U(-0.1,0.1); BN running_var ~ U(0.5,1.5)."""
    import zlib
    with torch.no_grad():
        for name, t in sorted(model.state_dict().items()):
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
            if not t.dtype.is_floating_point:
                t.zero_()
                continue
            u = torch.rand(t.shape, generator=g, dtype=torch.float32)
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                v = 0.5 + u
            elif leaf == "running_mean" or leaf == "bias" or "token" in name:
                v = (u - 0.5) * 0.2
            elif t.dim() == 1:
                v = 0.8 + 0.4 * u
            else:
                fan_in = t[0].numel()
                v = (u * 2 - 1) * math.sqrt(3.0 / fan_in)
            t.copy_(v.to(t.dtype))
    return model


# ----------------------------------------------------------------------------- engine pieces
def scale_and_translate_(pc, scale, shift):
    """PointcloudScaleAndTranslate with the draws injected (P/datasets/data_transforms.py:27-35):
    pc[i] = pc[i] * scale[i] + shift[i], in place.  scale ~ U[2/3,3/2]^3, shift ~ U[-0.2,0.2]^3."""
    pc[:, :, 0:3] = pc[:, :, 0:3] * scale.unsqueeze(1) + shift.unsqueeze(1)
    return pc


def adjust_learning_rate(epoch, lr, min_lr, warmup_epochs, epochs):  # P/util/lr_sched.py:11-23
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def param_groups(model, weight_decay):  # P/tools/builder.py:40-56
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.dim() == 1 or name.endswith(".bias") or "token" in name) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class ModelEma:  # timm 0.4.5 ModelEma: deepcopy().eval(), every state-dict entry lerped
    def __init__(self, model, decay=0.999):
        self.ema = copy.deepcopy(model).eval()
        self.decay = decay
        for p in self.ema.parameters():
            p.requires_grad_(False)

    @torch.no_grad()
    def update(self, model):
        msd = model.state_dict()
        for k, v in self.ema.state_dict().items():
            v.copy_(v * self.decay + (1.0 - self.decay) * msd[k].detach())


def ema_decay_for_epoch(epoch):  # P/engine_pretrain.py:55-60
    return 0.999 + epoch / 100 * (0.9999 - 0.999) if epoch < 100 else 0.9999


def pretrain_step(model, ema, optimizer, samples, epoch, total_epoch, mask_ratio=0.6, clip_grad=5.0,
                  mask_rng=None, mask_noise=None, relative=True):
    """One iteration of P/engine_pretrain.py:77-212 in fp32 (no autocast/GradScaler: the scaler is a
    no-op in fp32).  `samples` must already be augmented.  Returns a dict of step results."""
    B = samples.shape[0]
    visible = torch.zeros(B, model.num_group, dtype=torch.bool)
    with torch.no_grad():
        outs_ema = ema.ema(samples, mask=visible)
    mask = ema.ema.generate_mask(outs_ema["loss_pred"], mask_ratio=mask_ratio, guide=True, epoch=epoch,
                                 total_epoch=total_epoch, rng=mask_rng, noise=mask_noise).flatten(1).to(torch.bool)
    outs = model(samples, mask=mask)
    M = outs["mask_num"]
    loss_outs = model.forward_loss(outs["pix_pred"][:, -M:], outs["neighborhood"], outs["mask"])
    loss = 13.889 * loss_outs["MSE_mean"] + 1.0 * loss_outs["Chamfer_mean"]
    loss_learn = model.forward_learning_loss(outs["loss_pred"][:, -M:], mask, loss_outs["matrix"].detach(),
                                             relative=relative)
    total = loss + loss_learn
    optimizer.zero_grad()
    total.backward()
    grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
    optimizer.step()
    ema.update(model)
    return {"loss": total.detach(), "chamfer": loss_outs["Chamfer_mean"].detach(), "loss_learn": loss_learn.detach(),
            "grad_norm": grad_norm, "mask": mask, "teacher_loss_pred": outs_ema["loss_pred"],
            "matrix": loss_outs["matrix"].detach(), "outs": outs}
