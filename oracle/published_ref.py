"""CPU oracle of the published-run variant: plain PyTorch fp32 + oracle/ops.py.

TEST INFRASTRUCTURE ONLY (imported by tests/ only, never by gm3d_amd/).  Restates
P/models_mae_learn_loss_Classifier_SVM_feature_besed.py (MaskedAutoencoderViT :849-1059, generate_mask :1061-1110,
MaskTransformer :1329-1371), the frozen teacher P/models/Point_MAE.py (MaskTransformer :216-337 with mask_ratio 0,
Point_MAE :340-390) and one iteration of P/engine_pretrain_Classifier_SVM.py:98-290 with forward_features_dino_decoder
(:669-687).  P/ = /root/reference/Point-MAE_SA3D/.  Pinned against those files by tests/golden/make_golden_published.py
(this container only) -> tests/golden/published_b4.npz.  FPS / KNN / Chamfer stay "parity unpinned".
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import model_ref as R
from . import ops


def _pos_mlp(dim=384):
    return nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, dim))


class MaskTransformer(nn.Module):                                  # variant :1329-1371
    def __init__(self, drop_path_rate=0.1):
        super().__init__()
        self.encoder = R.Encoder(384)
        self.pos_embed = _pos_mlp()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 12)]
        self.blocks = R.TransformerEncoder(384, 12, 6, dpr)
        self.norm_p = nn.LayerNorm(384)

    def forward(self, neighborhood, center, mask):
        tokens = self.encoder(neighborhood)
        B, _, C = tokens.shape
        x_vis = tokens[~mask].reshape(B, -1, C)
        pos = self.pos_embed(center[~mask].reshape(B, -1, 3))
        return self.norm_p(self.blocks(x_vis, pos))


class PublishedGM3D(nn.Module):                                    # variant :849-1059
    def __init__(self, drop_path_rate=0.1):
        super().__init__()
        self.num_group, self.group_size, self.trans_dim = 64, 32, 384
        dpr4 = [x.item() for x in torch.linspace(0, drop_path_rate, 4)]
        dpr12 = [x.item() for x in torch.linspace(0, drop_path_rate, 12)]
        self.MAE_encoder = MaskTransformer(drop_path_rate)
        self.MAE_decoder = R.TransformerDecoder(384, 4, 6, dpr4)
        self.MAE_decoder_loss_pred = R.TransformerDecoder(384, 12, 6, dpr12)
        self.norm_p = nn.LayerNorm(384)
        self.group_divider = R.Group(64, 32)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, 384))
        self.mask_token_loss_pred = nn.Parameter(torch.zeros(1, 1, 384))
        self.decoder_pos_embed = _pos_mlp()
        self.increase_dim_2 = nn.Sequential(nn.Conv1d(384, 1024, 1), nn.BatchNorm1d(1024), nn.LeakyReLU(negative_slope=0.2),
                                            nn.Conv1d(1024, 384, 1))
        self.increase_dim_just_network_without_feature = nn.Sequential(nn.Conv1d(384, 96, 1))
        self.loss_func = ops.ChamferDistanceL2()

    def forward(self, pts, mask, shared_learnable_tokens=False, noaug=False):     # :1000-1059
        neighborhood, center, neighborhood_org = self.group_divider(pts)
        x_vis = self.MAE_encoder(neighborhood, center, mask)
        B, _, C = x_vis.shape
        if noaug:
            return x_vis
        pos_vis = self.decoder_pos_embed(center[~mask]).reshape(B, -1, C)
        pos_mask = self.decoder_pos_embed(center[mask]).reshape(B, -1, C)
        N = pos_mask.shape[1]
        x_full = torch.cat([x_vis, self.mask_token.expand(B, N, -1)], dim=1)
        pos_full = torch.cat([pos_vis, pos_mask], dim=1)
        loss_in = x_full.clone() if shared_learnable_tokens else torch.cat([x_vis, self.mask_token_loss_pred.expand(B, N, -1)], dim=1)
        x_rec = self.MAE_decoder(x_full, pos_full, N)
        lp = self.MAE_decoder_loss_pred(loss_in, pos_full, N)
        lp = self.increase_dim_2(lp.transpose(1, 2)).transpose(1, 2)
        return {"pix_pred": x_rec, "mask": mask, "mask_num": N, "features": x_vis, "loss_pred": lp.mean(dim=-1),
                "neighborhood": neighborhood, "neighborhood_org": neighborhood_org, "center": center}

    def forward_loss(self, pred, target, mask, point_target, point_reconstructed):      # :966-996
        N, P, D = target.shape
        target = target[mask].reshape(N, -1, D)
        PP = target.shape[1]
        pred = F.normalize(pred, p=2, dim=-1)
        target = F.normalize(target, p=2, dim=-1)
        loss_mse = ((pred - target) ** 2).sum(dim=-1)
        pt = point_target[mask].reshape(N * PP, -1, 3).float()
        pr = point_reconstructed.reshape(N * PP, -1, 3).float()
        loss_chamfer = self.loss_func(pr, pt).reshape(N, PP, -1).mean(-1)
        return {"MSE_mean": loss_mse.mean(), "Chamfer_mean": loss_chamfer.mean(), "matrix": loss_mse + loss_chamfer}

    @torch.no_grad()
    def generate_mask(self, loss_pred, mask_ratio=0.75, guide=True, epoch=0, total_epoch=200, after_200_epoch=None,
                      rng=None, noise=None):                                              # :1061-1110
        N, L = loss_pred.shape
        len_keep = int(L * (1 - mask_ratio))
        ids_loss = torch.argsort(loss_pred, dim=1)
        keep_ratio = 0.5
        if guide:
            keep_ratio = min(float((epoch + 1) / (total_epoch / 2)) * 0.5, 0.5) if after_200_epoch else float((epoch + 1) / total_epoch) * 0.8
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

    forward_learning_loss = R.PointMAEGM3D.forward_learning_loss                          # :1112-1131, same formula


class FrozenPointMAE(nn.Module):
    """P/models/Point_MAE.py::Point_MAE with config_m.yaml (mask_ratio 0): only what forward_features_dino_decoder touches."""

    class _Enc(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = R.Encoder(384)
            self.pos_embed = _pos_mlp()
            dpr = [x.item() for x in torch.linspace(0, 0.1, 12)]
            self.blocks = R.TransformerEncoder(384, 12, 6, dpr)
            self.norm = nn.LayerNorm(384)

        def forward(self, neighborhood, center):                     # :319-337 with an all-False mask
            tokens = self.encoder(neighborhood)
            return self.norm(self.blocks(tokens, self.pos_embed(center))), torch.zeros(center.shape[:2], dtype=torch.bool)

    def __init__(self):
        super().__init__()
        self.MAE_encoder = self._Enc()
        self.mask_token = nn.Parameter(torch.zeros(1, 1, 384))
        self.decoder_pos_embed = _pos_mlp()
        dpr = [x.item() for x in torch.linspace(0, 0.1, 4)]
        self.MAE_decoder = R.TransformerDecoder(384, 4, 6, dpr)
        self.increase_dim = nn.Sequential(nn.Conv1d(384, 96, 1))

    @torch.no_grad()
    def features_decoder(self, neighborhood, center, features, mask_real):               # engine :669-687
        x_vis, mask = self.MAE_encoder(neighborhood, center)
        B, N, C = x_vis.shape
        pos = self.decoder_pos_embed(center[~mask]).reshape(B, -1, C)
        pts_org = self.increase_dim(self.MAE_decoder(x_vis, pos, N).transpose(1, 2)).transpose(1, 2)
        pos = self.decoder_pos_embed(center[mask_real]).reshape(B, -1, C)
        pts_rec = self.increase_dim(self.MAE_decoder(features, pos, N).transpose(1, 2)).transpose(1, 2)
        return x_vis, pts_org, pts_rec


def pretrain_step(model, ema, teacher, optimizer, samples, epoch, total_epoch, mask_ratio=0.6, clip_grad=5.0, mask_rng=None,
                  mask_noise=None, after_epoch=15, loss_multiply_by=(13.889, 1000.0), relative=True):
    """One iteration of P/engine_pretrain_Classifier_SVM.py:98-290 in fp32; `samples` already augmented."""
    B = samples.shape[0]
    visible = torch.zeros(B, 64, dtype=torch.bool)
    with torch.no_grad():
        outs_ema = ema.ema(samples, mask=visible)
    mask = ema.ema.generate_mask(outs_ema["loss_pred"], mask_ratio=mask_ratio, guide=True, epoch=epoch, total_epoch=total_epoch,
                                 rng=mask_rng, noise=mask_noise).flatten(1).to(torch.bool)
    outs = model(samples, mask=mask)
    M = outs["mask_num"]
    with torch.no_grad():
        ft, pt, pr = teacher.features_decoder(outs_ema["neighborhood"], outs_ema["center"], outs["pix_pred"][:, -M:], outs["mask"])
    lo = model.forward_loss(outs["pix_pred"][:, -M:], ft.detach(), outs["mask"], pt, pr)
    w = (1.0, 1.0) if epoch < after_epoch else loss_multiply_by
    loss = w[0] * lo["MSE_mean"] + w[1] * lo["Chamfer_mean"]
    loss_learn = model.forward_learning_loss(outs["loss_pred"][:, -M:], mask, lo["matrix"].detach(), relative=relative)
    optimizer.zero_grad()
    (loss + loss_learn).backward()
    grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)
    optimizer.step()
    ema.update(model)
    return {"loss": loss.detach(), "mse": lo["MSE_mean"].detach(), "chamfer": lo["Chamfer_mean"].detach(),
            "loss_learn": loss_learn.detach(), "grad_norm": grad_norm, "mask": mask, "matrix": lo["matrix"].detach(),
            "teacher_loss_pred": outs_ema["loss_pred"]}
