"""Point_MAE -- the plain Point-MAE pre-training model of the reference (Point-MAE_SA3D/models/Point_MAE.py:216-441), used by
the published GM3D run as the FROZEN feature/point teacher (P/main_pretrain.py:302-328, config_m.yaml: mask_ratio 0).
Same constructor (`config` with group_size, num_group, loss, transformer_config.{mask_ratio, mask_type, trans_dim,
encoder_dims, depth, drop_path_rate, num_heads, decoder_depth, decoder_num_heads}) and state-dict keys
(MAE_encoder.{encoder,pos_embed,blocks,norm}.*, mask_token, decoder_pos_embed.*, MAE_decoder.*, increase_dim.*).
"""
import numpy as np
import torch
import torch.nn as nn

from . import models_mae_learn_loss as M
from .models_mae_learn_loss import Encoder, Group, TransformerDecoder, TransformerEncoder, split_ids, take
from .models_mae_learn_loss_Classifier_SVM_feature_besed import _pos_mlp
from .ops import ChamferDistanceL1, ChamferDistanceL2


def _get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


class MaskTransformer(nn.Module):
    """P/models/Point_MAE.py:216-337.  forward -> (x_vis, bool mask)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        tc = _get(config, "transformer_config")
        self.mask_ratio, self.mask_type = _get(tc, "mask_ratio"), _get(tc, "mask_type")
        self.trans_dim, self.depth = _get(tc, "trans_dim"), _get(tc, "depth")
        self.drop_path_rate, self.num_heads = _get(tc, "drop_path_rate"), _get(tc, "num_heads")
        self.encoder_dims = _get(tc, "encoder_dims")
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads)
        self.norm = nn.LayerNorm(self.trans_dim)

    def _mask_center_rand(self, center, noaug=False, rng=np.random):
        """P/:296-317: exactly int(mask_ratio*G) masked groups per cloud, uniformly at random (host shuffle like the reference)."""
        B, G, _ = center.shape
        if noaug or self.mask_ratio == 0:
            return torch.zeros(B, G, dtype=torch.bool, device=center.device)
        num_mask = int(self.mask_ratio * G)
        out = np.zeros([B, G])
        for i in range(B):
            m = np.hstack([np.zeros(G - num_mask), np.ones(num_mask)])
            rng.shuffle(m)
            out[i] = m
        return torch.from_numpy(out).to(torch.bool).to(center.device)

    def _mask_center_block(self, center, noaug=False, rng=np.random):
        """P/:268-294: the int(mask_ratio*G) groups nearest to a random centre."""
        B, G, _ = center.shape
        if noaug or self.mask_ratio == 0:
            return torch.zeros(B, G, dtype=torch.bool, device=center.device)
        start = torch.as_tensor(rng.randint(0, G, size=B), device=center.device)
        d = torch.norm(center[torch.arange(B, device=center.device), start].unsqueeze(1) - center, p=2, dim=-1)
        idx = torch.argsort(d, dim=-1)[:, :int(self.mask_ratio * G)]
        mask = torch.zeros(B, G, dtype=torch.bool, device=center.device)
        mask.scatter_(1, idx, True)
        return mask

    def forward(self, neighborhood, center, noaug=False):
        mask = (self._mask_center_rand if self.mask_type == "rand" else self._mask_center_block)(center, noaug=noaug)
        G = center.shape[1]
        num_vis = G if (noaug or self.mask_ratio == 0) else G - int(self.mask_ratio * G)
        vis_ids, _ = split_ids(mask, num_vis)
        tokens = self.encoder(neighborhood)
        pos = _pos_mlp(self.pos_embed, center)
        return self.blocks(take(tokens, vis_ids), take(pos, vis_ids), norm=self.norm), mask


class Point_MAE(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        tc = _get(config, "transformer_config")
        self.trans_dim = _get(tc, "trans_dim")
        self.MAE_encoder = MaskTransformer(config)
        self.group_size, self.num_group = _get(config, "group_size"), _get(config, "num_group")
        self.drop_path_rate = _get(tc, "drop_path_rate")
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.decoder_pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        self.decoder_depth, self.decoder_num_heads = _get(tc, "decoder_depth"), _get(tc, "decoder_num_heads")
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.decoder_depth)]
        self.MAE_decoder = TransformerDecoder(embed_dim=self.trans_dim, depth=self.decoder_depth, drop_path_rate=dpr,
                                              num_heads=self.decoder_num_heads)
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.increase_dim = nn.Sequential(nn.Conv1d(self.trans_dim, 3 * self.group_size, 1))
        nn.init.trunc_normal_(self.mask_token, std=0.02)
        self.loss = _get(config, "loss")
        self.build_loss_func(self.loss)

    def build_loss_func(self, loss_type):
        if loss_type == "cdl1":
            self.loss_func = ChamferDistanceL1()
        elif loss_type == "cdl2":
            self.loss_func = ChamferDistanceL2(reduction="mean")
        else:
            raise NotImplementedError

    def _points(self, x_rec):
        c = self.increase_dim[0]
        if M.FUSED_HEADS and x_rec.is_cuda:
            from . import heads
            return heads.LinearBiasFn.apply(x_rec, c.weight, c.bias, heads._adt())
        return nn.functional.linear(x_rec, c.weight.squeeze(-1), c.bias)

    def forward(self, pts, noaug=False, vis=False, **kwargs):
        """P/:391-441 (Point-MAE pre-training loss; `vis` is a plotting aid and is not provided)."""
        neighborhood, center, _ = self.group_divider(pts)
        x_vis, mask = self.MAE_encoder(neighborhood, center, noaug=noaug)
        B, V, C = x_vis.shape
        if noaug:
            return x_vis
        vis_ids, mask_ids = split_ids(mask, V)
        dpos = _pos_mlp(self.decoder_pos_embed, center)
        N = mask_ids.shape[1]
        x_full = torch.cat([x_vis, self.mask_token.expand(B, N, -1).to(x_vis.dtype)], dim=1)
        pos_full = torch.cat([take(dpos, vis_ids), take(dpos, mask_ids)], dim=1)
        x_rec = self.MAE_decoder(x_full, pos_full, N)[:, -N:]          # P/:212: only the masked tokens are decoded to points
        rebuild = self._points(x_rec).reshape(B * N, -1, 3).float()
        gt = take(neighborhood, mask_ids).reshape(B * N, -1, 3)
        return self.loss_func(rebuild, gt)

    @torch.no_grad()
    def features_decoder(self, neighborhood, center, features, mask_ids):
        """forward_features_dino_decoder (P/engine_pretrain_Classifier_SVM.py:669-687) for the all-visible teacher:
        -> (features of all G tokens (B,G,C), their decoded points (B,G,3k), points decoded from the student's `features`
        (B,M,C) placed at the masked centres (B,M,3k))."""
        x_vis, _ = self.MAE_encoder(neighborhood, center, noaug=True)
        dpos = _pos_mlp(self.decoder_pos_embed, center)
        G = x_vis.shape[1]
        pts_org = self._points(self.MAE_decoder(x_vis, dpos, G))
        pts_rec = self._points(self.MAE_decoder(features, take(dpos, mask_ids), G))
        return x_vis, pts_org, pts_rec
