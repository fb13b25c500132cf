"""Checkpoint format of the reference's training scripts, written and read without pickled code.

Layout (Point-MAE_SA3D/main_pretrain_multi_gpu.py:355-385, main_finetune.py:411-450):
    {"epoch": epoch + 1, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(), "model": <model name>,
     "ema_state_dict": model_ema.ema.state_dict() (when a teacher exists), "loss_scaler": ... (when AMP scaling is used)}
Resume follows util/misc.py:317-342 (load_model); fine-tune initialisation follows main_finetune.py:296-325
(key 'ema_state_dict' for --teacher, else 'state_dict' / 'model'; 'module.' and 'MAE_encoder.' prefixes dropped;
strict=False).

The 'optimizer' entry: torch.optim.AdamW's state_dict when the run uses it; with FlatAdamWEma (the measured configuration) it is
that optimizer's own layout plus `param_names` -- restored by parameter name into any FlatAdamWEma over the same model, but NOT
loadable by the reference's optimizer.load_state_dict (util/misc.py:335: two groups [no_decay, decay] in named_parameters order
over all 470 tensors, dead ones included).  Model and EMA tensors are interchangeable with the reference in both directions.

Files hold tensors, numbers, strings, lists and dicts only, so they load with torch.load(weights_only=True) -- here and
in the reference (whose torch.load default accepts them as well).  The reference model additionally carries ~230 dead
image-MAE entries (patch_embed.*, decoder_*, ...: SURVEY.md 0.7); `reference_keys=` lets a caller add them (zeros) so
that the reference's strict resume path accepts the file.
"""
import os

import torch


def _cpu(sd):
    return {k: (v.detach().to("cpu").clone() if torch.is_tensor(v) else v) for k, v in sd.items()}


def _plain(obj):
    """optimizer.state_dict() -> CPU tensors / python scalars only."""
    if torch.is_tensor(obj):
        return obj.detach().to("cpu").clone()
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


def checkpoint_dict(model, optimizer=None, epoch=0, model_name="", model_ema=None, loss_scaler=None, reference_keys=None):
    net = model.module if hasattr(model, "module") else model
    sd = _cpu(net.state_dict())
    if reference_keys:                      # {key: shape} of the reference model: add its dead entries as zeros
        for k, shape in reference_keys.items():
            if k not in sd:
                sd[k] = torch.zeros(tuple(shape), dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    out = {"epoch": epoch + 1, "state_dict": sd, "model": model_name}
    if optimizer is not None:
        out["optimizer"] = _plain(optimizer.state_dict())
    if model_ema is not None:
        out["ema_state_dict"] = _cpu(model_ema.ema.state_dict())
    if loss_scaler is not None:
        out["loss_scaler"] = _plain(loss_scaler.state_dict())
    return out


def save_checkpoint(path, model, optimizer=None, epoch=0, model_name="", model_ema=None, loss_scaler=None,
                    reference_keys=None, is_master=True):
    """utils.save_on_master(save_dict, ckpt_path) of the reference; written atomically."""
    if not is_master:
        return None
    d = checkpoint_dict(model, optimizer, epoch, model_name, model_ema, loss_scaler, reference_keys)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = path + ".tmp"
    torch.save(d, tmp)
    os.replace(tmp, path)
    return path


def read_checkpoint(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def load_checkpoint(path, model, optimizer=None, loss_scaler=None, model_ema=None, evaluate_only=False):
    """Resume (P/util/misc.py:317-342): model <- 'state_dict' (strict), teacher <- 'ema_state_dict', optimizer and
    loss scaler when present.  -> start epoch (the stored 'epoch', as the reference sets args.start_epoch) or None."""
    ckpt = read_checkpoint(path)
    net = model.module if hasattr(model, "module") else model
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt["module"]
    if hasattr(net, "load_reference_state_dict"):
        net.load_reference_state_dict(sd)          # strict on the live keys, ignores the reference's dead ones
    else:
        net.load_state_dict(sd, strict=True)
    if "ema_state_dict" in ckpt and model_ema is not None:
        ema = model_ema.ema
        if hasattr(ema, "load_reference_state_dict"):
            ema.load_reference_state_dict(ckpt["ema_state_dict"])
        else:
            ema.load_state_dict(ckpt["ema_state_dict"])
    if optimizer is not None and hasattr(optimizer, "sync_shadows"):
        optimizer.sync_shadows()                   # flat optimizer: bf16 GEMM copies follow the restored masters
    start = None
    if optimizer is not None and "optimizer" in ckpt and "epoch" in ckpt and not evaluate_only:
        optimizer.load_state_dict(ckpt["optimizer"])
        start = ckpt["epoch"]
        if loss_scaler is not None and "loss_scaler" in ckpt:
            loss_scaler.load_state_dict(ckpt["loss_scaler"])
    return start


def finetune_state(ckpt, teacher=False):
    """The state dict main_finetune.py:300-312 hands to load_state_dict(strict=False)."""
    from .point_transformer import strip_pretrain_prefixes
    key = "ema_state_dict" if teacher else ("state_dict" if "state_dict" in ckpt else "model")
    return strip_pretrain_prefixes(ckpt[key])
