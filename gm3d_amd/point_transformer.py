"""PointTransformer -- the fine-tune classifier of the reference, on the same HIP kernels as the
pretrain path: FPS + fused KNN grouping, the mini-PointNet token embed node, the fused 12-block transformer stack
(attention tiles for T = 1 cls + 64 patch tokens) and the positional-embed node.

Mirrors Point-MAE_SA3D/models/Point_MAE.py:444-579 (class PointTransformer): same constructor (`config` with trans_dim,
depth, drop_path_rate, cls_dim, num_heads, group_size, num_group, encoder_dims), same state-dict keys
(group_divider has no state; encoder.*, cls_token, cls_pos, pos_embed.*, blocks.blocks.{i}.*, norm_p.*,
cls_head_finetune.{0,1,4,5,8}.*), same forward / get_loss_acc / load_model_from_ckpt behaviour.
"""
import torch
import torch.nn as nn

from .models_mae_learn_loss import Encoder, Group, TransformerEncoder, FUSED_HEADS, linear3  # noqa: F401
from . import models_mae_learn_loss as M


def _cfg(config, name):
    return config[name] if isinstance(config, dict) else getattr(config, name)


def strip_pretrain_prefixes(state_dict):
    """The key rewriting both loaders of the reference apply to a pre-training checkpoint before load_state_dict(strict=False):
    drop DDP's 'module.', then 'MAE_encoder.' / 'base_model.' (P/models/Point_MAE.py:509-517, P/main_finetune.py:311-312)."""
    out = {}
    for k, v in state_dict.items():
        k = k.replace("module.", "")
        if k.startswith("MAE_encoder."):
            k = k[len("MAE_encoder."):]
        elif k.startswith("base_model."):
            k = k[len("base_model."):]
        out[k] = v
    return out


class PointTransformer(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        self.trans_dim = _cfg(config, "trans_dim")
        self.depth = _cfg(config, "depth")
        self.drop_path_rate = _cfg(config, "drop_path_rate")
        self.cls_dim = _cfg(config, "cls_dim")
        self.num_heads = _cfg(config, "num_heads")
        self.group_size = _cfg(config, "group_size")
        self.num_group = _cfg(config, "num_group")
        self.encoder_dims = _cfg(config, "encoder_dims")
        if self.num_group + 1 > 512:
            raise NotImplementedError("the HIP attention kernels hold at most 512 tokens (cls + num_group)")

        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.trans_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                         num_heads=self.num_heads)
        self.norm_p = nn.LayerNorm(self.trans_dim)
        self.cls_head_finetune = nn.Sequential(
            nn.Linear(self.trans_dim * 2, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
            nn.Linear(256, self.cls_dim))
        self.build_loss_func()
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        nn.init.trunc_normal_(self.cls_pos, std=0.02)

    def build_loss_func(self):
        self.loss_ce = nn.CrossEntropyLoss()

    def get_loss_acc(self, ret, gt):
        loss = self.loss_ce(ret, gt.long())
        pred = ret.argmax(-1)
        acc = (pred == gt).sum() / float(gt.size(0))
        return loss, acc * 100

    # ------------------------------------------------------------------ checkpoints
    def load_pretrained_state(self, state_dict):
        """state_dict: a pre-training checkpoint's 'state_dict' / 'model' / 'ema_state_dict' / 'base_model' entry.
        -> the `_IncompatibleKeys` of load_state_dict(strict=False), like P/main_finetune.py:324."""
        return self.load_state_dict(strip_pretrain_prefixes(state_dict), strict=False)

    def load_model_from_ckpt(self, bert_ckpt_path, key="base_model"):
        """P/models/Point_MAE.py:506-540.  The file is read with torch.load(weights_only=True): checkpoints written by
        gm3d_amd.checkpoint hold tensors and plain containers only."""
        if bert_ckpt_path is None:
            self.apply(self._init_weights)
            return None
        ckpt = torch.load(bert_ckpt_path, map_location="cpu", weights_only=True)
        if key not in ckpt:
            key = next(k for k in ("base_model", "state_dict", "model") if k in ckpt)
        return self.load_pretrained_state(ckpt[key])

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ forward
    def embed_pos(self, center):
        l0, act, l1 = self.pos_embed
        if M.FUSED_HEADS and center.is_cuda:
            from . import heads
            return heads.PosEmbedFn.apply(center, l0.weight, l0.bias, l1.weight, l1.bias, heads._adt())
        return l1(act(linear3(center, l0.weight, l0.bias)))

    def forward_features(self, pts):
        """-> (B, 2*trans_dim): [cls token ; max over the patch tokens] after the 12 blocks and norm_p (P/:565-573)."""
        neighborhood, center, _ = self.group_divider(pts)
        tokens = self.encoder(neighborhood)                                   # B G C
        B = tokens.size(0)
        pos = self.embed_pos(center)
        x = torch.cat((self.cls_token.expand(B, -1, -1).to(tokens.dtype), tokens), dim=1)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1).to(pos.dtype), pos), dim=1)
        x = self.blocks(x, pos, norm=self.norm_p)                             # fused stack + norm_p
        return torch.cat([x[:, 0], x[:, 1:].max(1)[0]], dim=-1)

    def forward(self, pts):
        return self.cls_head_finetune(self.forward_features(pts))
