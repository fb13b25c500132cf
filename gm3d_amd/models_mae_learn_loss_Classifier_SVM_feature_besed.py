"""The published-run variant of the GM3D model on the same HIP kernels / fused autograd nodes as the
north-star model: separate `MAE_encoder` (MaskTransformer), 4-block reconstruction decoder, 12-block loss-prediction
decoder with its own mask token, `decoder_pos_embed`, feature-space MSE + Chamfer loss against a frozen Point-MAE teacher.

Mirrors Point-MAE_SA3D/models_mae_learn_loss_Classifier_SVM_feature_besed.py: MaskedAutoencoderViT (:849-1059),
generate_mask (:1061-1110), forward_learning_loss (:1112-1131), MaskTransformer (:1329-1371), factories (:1134-1158).
Same state-dict keys (MAE_encoder.{encoder,pos_embed,blocks,norm_p}.*, MAE_decoder.*, MAE_decoder_loss_pred.*, norm_p.*,
mask_token, mask_token_loss_pred, decoder_pos_embed.*, increase_dim_2.*, increase_dim_just_network_without_feature.*),
same forward dict.  `P/` = /root/reference/Point-MAE_SA3D/.

Execution differences (results unchanged): pos_embed / decoder_pos_embed are row-wise MLPs, so they are evaluated once on
all 64 centres and gathered (the reference evaluates them on the boolean-mask selections); boolean-mask gathers become
index gathers with a static visible count; the reconstruction head `increase_dim_just_network_without_feature` is not
evaluated in forward -- the reference computes it and discards the result (:1034-1035), its parameters never receive a
gradient there either.
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import models_mae_learn_loss as M
from .models_mae_learn_loss import Encoder, Group, TransformerDecoder, TransformerEncoder, split_ids, take
from .ops import ChamferDistanceL2


def _pos_mlp(seq, center):
    l0, act, l1 = seq
    if M.FUSED_HEADS and center.is_cuda:
        from . import heads
        return heads.PosEmbedFn.apply(center, l0.weight, l0.bias, l1.weight, l1.bias, heads._adt())
    return l1(act(M.linear3(center, l0.weight, l0.bias)))


class MaskTransformer(nn.Module):
    """P/:1329-1371: embed -> visible tokens -> pos_embed(visible centres) -> 12 blocks -> norm_p."""

    def __init__(self):
        super().__init__()
        self.encoder_dims = self.trans_dim = 384
        self.depth, self.drop_path_rate, self.num_heads = 12, 0.1, 6
        self.group_size, self.num_group = 32, 64
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, 384))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads)
        self.norm_p = nn.LayerNorm(self.trans_dim)

    def forward(self, neighborhood, center, mask, num_visible=None, ids=None):
        vis_ids, _ = ids if ids is not None else split_ids(mask, num_visible)
        pos = _pos_mlp(self.pos_embed, center)
        if M.VISIBLE_EMBED and vis_ids.shape[1] < self.num_group and self.encoder.fused(neighborhood):
            tok_vis = self.encoder(neighborhood, vis_ids=vis_ids)     # last embed conv on the visible groups only (embed.EmbedFn)
        else:
            tok_vis = take(self.encoder(neighborhood), vis_ids)
        return self.blocks(tok_vis, take(pos, vis_ids), norm=self.norm_p)


class MaskedAutoencoderViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0,
                 norm_layer=nn.LayerNorm, norm_pix_loss=False, asymmetric_decoder=False, mask_ratio=0.75,
                 vis_mask_ratio=0.0, saliency=False):
        super().__init__()
        self.encoder_dims = self.trans_dim = 384
        self.depth, self.drop_path_rate, self.num_heads = 12, 0.1, 6
        self.group_size, self.num_group = 32, 64
        self.decoder_depth, self.decoder_num_heads = 4, 6
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.decoder_depth)]
        self.MAE_encoder = MaskTransformer()
        self.MAE_decoder = TransformerDecoder(embed_dim=self.trans_dim, depth=self.decoder_depth, drop_path_rate=dpr,
                                              num_heads=self.decoder_num_heads)
        dpr12 = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.MAE_decoder_loss_pred = TransformerDecoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr12,
                                                        num_heads=self.decoder_num_heads)        # 12 blocks (P/:893-898)
        self.norm_p = nn.LayerNorm(self.trans_dim)                                            # unused by forward, like P/:901
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.mask_token_loss_pred = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.decoder_pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        self.increase_dim_2 = nn.Sequential(nn.Conv1d(self.trans_dim, 1024, 1), nn.BatchNorm1d(1024),
                                            nn.LeakyReLU(negative_slope=0.2), nn.Conv1d(1024, self.trans_dim, 1))
        self.increase_dim_just_network_without_feature = nn.Sequential(nn.Conv1d(self.trans_dim, 3 * self.group_size, 1))
        self.loss_func = ChamferDistanceL2()

    _loss_pred_head = M.MaskedAutoencoderViT._loss_pred_head
    forward_learning_loss = M.MaskedAutoencoderViT.forward_learning_loss          # P/:1112-1131, identical formula
    load_reference_state_dict = M.MaskedAutoencoderViT.load_reference_state_dict

    def _expand(self, token, B, N, dtype):
        if M.FUSED_HEADS and token.is_cuda:
            from . import heads
            return heads.ExpandRowsFn.apply(token, B, N, dtype)
        return token.expand(B, N, -1).to(dtype)

    def forward(self, pts, mask, shared_learnable_tokens=False, noaug=False, num_visible=None, group=None, ids=None):
        """P/:1000-1059.  Extra keyword-only conveniences for the engine as in the north-star model: `num_visible`, `group`
        (a previously computed grouping), `ids` (visible / masked id lists)."""
        neighborhood, center, neighborhood_org = group if group is not None else self.group_divider(pts)
        vis_ids, mask_ids = ids if ids is not None else split_ids(mask, num_visible)
        x_vis = self.MAE_encoder(neighborhood, center, mask, ids=(vis_ids, mask_ids))
        B, _, C = x_vis.shape
        if noaug:
            return x_vis
        N = mask_ids.shape[1]
        dpos = _pos_mlp(self.decoder_pos_embed, center)
        pos_full = torch.cat([take(dpos, vis_ids), take(dpos, mask_ids)], dim=1)
        x_full = torch.cat([x_vis, self._expand(self.mask_token, B, N, x_vis.dtype)], dim=1)
        if shared_learnable_tokens:
            loss_in = x_full
        else:
            loss_in = torch.cat([x_vis, self._expand(self.mask_token_loss_pred, B, N, x_vis.dtype)], dim=1)
        x_rec = self.MAE_decoder(x_full, pos_full, N)
        loss_pred_ = self.MAE_decoder_loss_pred(loss_in, pos_full, N)
        return {
            "pix_pred": x_rec,                                   # decoder FEATURES of all tokens (P/:1046)
            "mask": mask,
            "mask_num": N,
            "features": x_vis,
            "loss_pred": self._loss_pred_head(loss_pred_),
            "neighborhood": neighborhood,
            "neighborhood_org": neighborhood_org,
            "center": center,
        }

    def forward_loss(self, pred, target, mask, point_target, point_reconstructed, mask_ids=None):
        """P/:966-996.  pred (B,M,C) student features of the masked tokens; target (B,G,C) frozen-teacher features of all
        tokens; point_target (B,G,96) / point_reconstructed (B,M,96) the frozen teacher's decoded points.
        MSE on L2-normalised features + per-token Chamfer; `matrix` = their sum (the loss-predictor's target)."""
        N, P, D = target.shape
        if mask_ids is None:
            _, mask_ids = split_ids(mask, P - pred.shape[1])
        target = take(target, mask_ids)
        PP = target.shape[1]
        pred = F.normalize(pred.float(), p=2, dim=-1)
        target = F.normalize(target.float(), p=2, dim=-1)
        loss_mse = ((pred - target) ** 2).sum(dim=-1)
        pt = take(point_target, mask_ids).reshape(N * PP, -1, 3).to(torch.float32)
        pr = point_reconstructed.reshape(N * PP, -1, 3).to(torch.float32)
        loss_chamfer = self.loss_func(pr, pt).reshape(N, PP, -1).mean(-1)
        return {"MSE_mean": loss_mse.mean(), "Chamfer_mean": loss_chamfer.mean(), "matrix": loss_mse + loss_chamfer}

    @staticmethod
    def keep_ratio(guide, epoch, total_epoch, after_200_epoch):
        """P/:1069-1074."""
        if not guide:
            return 0.5
        if after_200_epoch:
            return min(float((epoch + 1) / (total_epoch / 2)) * 0.5, 0.5)
        return float((epoch + 1) / total_epoch) * 0.8

    @torch.no_grad()
    def generate_mask_ids(self, loss_pred, mask_ratio=0.75, guide=True, epoch=0, total_epoch=200, after_200_epoch=None,
                          noise=None):
        """generate_mask (P/:1061-1110) + the id lists, one launch (gm3d_mask_select); same construction as the north-star
        model's with this variant's keep_ratio schedule."""
        from ._capi import lib
        from . import ops
        N, L = loss_pred.shape
        len_keep = int(L * (1 - mask_ratio))
        len_loss = max(int((L - len_keep) * self.keep_ratio(guide, epoch, total_epoch, after_200_epoch)), 0)
        dev = loss_pred.device
        noise = torch.rand(N, L, device=dev) if noise is None else noise.to(dev, torch.float32).contiguous()
        lp = loss_pred.detach().float().contiguous()
        mask = torch.empty(N, L, dtype=torch.float32, device=dev)
        order = torch.empty(N, L, dtype=torch.int64, device=dev)        # [visible ids | masked ids] per sample
        vis_ids, mask_ids = order[:, :len_keep], order[:, len_keep:]
        ops._launch("gm3d_mask_select", {"B": N, "L": L}, lib.gm3d_mask_select, ops._ptr(lp), ops._ptr(noise), N, L, len_keep,
                    len_loss, ops._ptr(mask), ops._ptr(vis_ids), ops._ptr(mask_ids), L, ops._stream())
        return mask, vis_ids, mask_ids

    @torch.no_grad()
    def generate_mask(self, loss_pred, mask_ratio=0.75, images=None, guide=True, epoch=0, total_epoch=200,
                      after_200_epoch=None, noise=None):
        return self.generate_mask_ids(loss_pred, mask_ratio, guide, epoch, total_epoch, after_200_epoch, noise)[0]


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
