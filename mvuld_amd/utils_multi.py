"""Checkpoint / resume / reduce helpers with the reference's names, file layout and dict keys
(mvuld/utils_multi.py): ``load_checkpoint`` :7-32, ``save_checkpoint`` :125-137, ``save_bestf1_checkpoint`` :139-152,
``resume_bestf1_helper`` :154-168, ``auto_resume_helper`` :183-195, ``reduce_tensor`` :198-202,
``NativeScalerWithGradNormCount`` :220-246 (the GradScaler is vestigial there: autocast is disabled around the model
call, main_bigvul.py:328 -- here the step is bf16/fp32 without loss scaling, the clip + AdamW run as fused kernels)."""
import os

import torch
import torch.distributed as dist


def _after_load(model):
    store = getattr(model, "_mv_store", None)
    if store is not None:
        store.refresh_working_copy()
    else:
        from . import ops
        ops.bump_weight_epoch()


_GEOMETRY_KEYS = ("relative_position_index", "relative_coords_table", "attn_mask")


def remap_reference_keys(state_dict, model):
    """Keys of a checkpoint written by the reference, renamed for `model`.  The reference trains and saves the bare fusion head
    (Multi_DefectModel_new_GCN: `gat.fc.weight`, `Rs_GCN_1.g.weight`, ...; main_bigvul.py:258-262, utils_multi.py:139-152); the
    fused model here holds the same module under `head.` (and the encoders under `swin.` / `unixcoder.`).  Keys that already
    resolve are left alone."""
    own = set(model.state_dict().keys())
    if not own or any(k in own for k in state_dict):
        return dict(state_dict)
    for prefix in ("head.", "swin.", "unixcoder."):
        if any(prefix + k in own for k in state_dict):
            return {prefix + k: v for k, v in state_dict.items()}
    return dict(state_dict)


def load_state_dict_checked(model, state_dict, logger=None, what="checkpoint"):
    """load_state_dict(strict=False) that says what it did: logs missing / unexpected keys and refuses a checkpoint none of whose
    keys belong to the model (the reference's strict=False load, utils_multi.py:14, would carry on from random init)."""
    state_dict = remap_reference_keys(state_dict, model)
    own = set(model.state_dict().keys())
    hit = [k for k in state_dict if k in own or any(k.endswith(g) for g in _GEOMETRY_KEYS)]
    if not hit:
        raise RuntimeError(f"{what}: none of its {len(state_dict)} keys (e.g. {list(state_dict)[:3]}) matches the model "
                           f"(e.g. {sorted(own)[:3]}): wrong checkpoint for this model")
    msg = model.load_state_dict(state_dict, strict=False)
    if logger is not None:
        if msg.missing_keys:
            logger.warning(f"{what}: {len(msg.missing_keys)} model keys not in the file (kept as initialised), e.g. {msg.missing_keys[:5]}")
        if msg.unexpected_keys:
            logger.warning(f"{what}: {len(msg.unexpected_keys)} file keys not in the model (ignored), e.g. {msg.unexpected_keys[:5]}")
    return msg


def load_checkpoint(config, model, optimizer, lr_scheduler, loss_scaler, logger, path=None):
    """``path``: the checkpoint file; default ``config.MODEL.MULTI.RESUME`` (the multimodal job, utils_multi.py:15-33); the Swin fine-tune
    job passes ``config.MODEL.RESUME`` (utils.py load_checkpoint as called at main.py:162,179)."""
    path = path or config.MODEL.MULTI.RESUME
    logger.info(f"==============> Resuming form {path}....................")
    if path.startswith('https'):
        raise RuntimeError("no network on this box: pass a local checkpoint path")
    checkpoint = torch.load(path, map_location='cpu', weights_only=False)
    load_state_dict_checked(model, checkpoint['model'], logger, path)
    _after_load(model)
    max_accuracy = 0.0
    epoch = checkpoint['epoch']
    if not config.EVAL_MODE and 'optimizer' in checkpoint and 'lr_scheduler' in checkpoint and 'epoch' in checkpoint:
        optimizer.load_state_dict(checkpoint['optimizer'])
        lr_scheduler.load_state_dict(checkpoint['lr_scheduler'])
        config.defrost()
        config.TRAIN.START_EPOCH = checkpoint['epoch'] + 1
        config.freeze()
        if 'scaler' in checkpoint:
            loss_scaler.load_state_dict(checkpoint['scaler'])
        logger.info(f"=> loaded successfully '{path}' (epoch {checkpoint['epoch']})")
        if 'max_accuracy' in checkpoint:
            max_accuracy = checkpoint['max_accuracy']
    del checkpoint
    return max_accuracy, epoch


def load_pretrained(config, model, logger):
    """Fine-tuning start from a (Swin) checkpoint `config.MODEL.PRETRAINED` = {'model': state_dict}, as the reference's
    load_pretrained (utils_multi.py:35-122): geometry buffers of the file (relative_position_index / relative_coords_table /
    attn_mask) are dropped -- this implementation derives them from the window geometry; a Swin-v1 style
    relative_position_bias_table of another window size is resized bicubically over its (2w-1)x(2w-1) grid; an
    absolute_pos_embed of another resolution likewise; a classifier head of another class count is re-initialised to zero."""
    import torch.nn.functional as F
    logger.info(f"==============> Loading weight {config.MODEL.PRETRAINED} for fine-tuning......")
    checkpoint = torch.load(config.MODEL.PRETRAINED, map_location='cpu', weights_only=False)
    sd = dict(checkpoint['model'] if 'model' in checkpoint else checkpoint)
    for k in [k for k in sd if any(g in k for g in _GEOMETRY_KEYS)]:
        del sd[k]
    own = model.state_dict()
    for k in [k for k in sd if "relative_position_bias_table" in k]:
        if k not in own:
            continue
        src, cur = sd[k], own[k]
        (L1, nH1), (L2, nH2) = src.shape, cur.shape
        if nH1 != nH2:
            logger.warning(f"Error in loading {k}, passing......")
            del sd[k]
        elif L1 != L2:
            S1, S2 = int(round(L1 ** 0.5)), int(round(L2 ** 0.5))
            grid = F.interpolate(src.t().reshape(1, nH1, S1, S1), size=(S2, S2), mode='bicubic')
            sd[k] = grid.reshape(nH2, L2).t().contiguous()
    for k in [k for k in sd if "absolute_pos_embed" in k]:
        if k not in own:
            continue
        src, cur = sd[k], own[k]
        (_, L1, C1), (_, L2, C2) = src.shape, cur.shape
        if C1 != C2:
            logger.warning(f"Error in loading {k}, passing......")
            del sd[k]
        elif L1 != L2:
            S1, S2 = int(round(L1 ** 0.5)), int(round(L2 ** 0.5))
            grid = F.interpolate(src.reshape(-1, S1, S1, C1).permute(0, 3, 1, 2), size=(S2, S2), mode='bicubic')
            sd[k] = grid.permute(0, 2, 3, 1).flatten(1, 2)
    head = getattr(model, "head", None)
    if 'head.bias' in sd and isinstance(head, torch.nn.Linear) and sd['head.bias'].shape[0] != head.bias.shape[0]:
        torch.nn.init.constant_(head.bias, 0.)
        torch.nn.init.constant_(head.weight, 0.)
        del sd['head.weight'], sd['head.bias']
        logger.warning("Error in loading classifier head, re-init classifier head to 0")
    msg = load_state_dict_checked(model, sd, logger, config.MODEL.PRETRAINED)
    _after_load(model)
    logger.info(f"=> loaded successfully '{config.MODEL.PRETRAINED}'")
    return msg


def load_fused_parts(model, swin_ckpt=None, unixcoder_bin=None, head_ckpt=None, logger=None):
    """Fill a FusedMVulD from the three files the reference's pipeline produces (data/bigvul_dataset.py:60-99, main_bigvul.py:258-262):
    the fine-tuned Swin `ckpt['model']` (its 2-class `head.*` is kept as is), the fine-tuned UniXcoder `pytorch_model.bin` (bare
    state dict of MyUniXcoder: `encoder.*`, `classifier.*`, separate query / key / value) and the fusion head's `mymodel.pth`."""
    out = {}
    if swin_ckpt is not None:
        sd = torch.load(swin_ckpt, map_location='cpu', weights_only=False)
        sd = dict(sd['model'] if 'model' in sd else sd)
        hb = model.swin.head.bias.shape[0] if isinstance(model.swin.head, torch.nn.Linear) else None
        if hb is not None and 'head.bias' in sd and sd['head.bias'].shape[0] != hb:        # bigvul_dataset.py:75-76
            del sd['head.weight'], sd['head.bias']
        out["swin"] = load_state_dict_checked(model.swin, sd, logger, swin_ckpt)
    if unixcoder_bin is not None:
        sd = torch.load(unixcoder_bin, map_location='cpu', weights_only=False)
        out["unixcoder"] = load_state_dict_checked(model.unixcoder, dict(sd), logger, unixcoder_bin)
    if head_ckpt is not None:
        sd = torch.load(head_ckpt, map_location='cpu', weights_only=False)
        sd = dict(sd['model'] if 'model' in sd else sd)
        out["head"] = load_state_dict_checked(model.head, sd, logger, head_ckpt)
    _after_load(model)
    return out


def _state(config, epoch, model, max_accuracy, optimizer, lr_scheduler, loss_scaler):
    return {'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'lr_scheduler': lr_scheduler.state_dict(),
            'max_accuracy': max_accuracy, 'scaler': loss_scaler.state_dict(), 'epoch': epoch,
            'config': config.to_dict() if hasattr(config, "to_dict") else config}


def save_checkpoint(config, epoch, model, max_accuracy, optimizer, lr_scheduler, loss_scaler, logger):
    save_path = os.path.join(config.MULTI_OUTPUT, f'ckpt_epoch_{epoch}.pth')
    logger.info(f"{save_path} saving......")
    torch.save(_state(config, epoch, model, max_accuracy, optimizer, lr_scheduler, loss_scaler), save_path)
    logger.info(f"{save_path} saved !!!")


def save_bestf1_checkpoint(config, epoch, model, max_accuracy, optimizer, lr_scheduler, loss_scaler, logger):
    out = os.path.join(config.MULTI_OUTPUT, 'checkpoint-best-f1')
    os.makedirs(out, exist_ok=True)
    save_path = os.path.join(out, 'mymodel.pth')
    logger.info(f"{save_path} best f1 saving......")
    torch.save(_state(config, epoch, model, max_accuracy, optimizer, lr_scheduler, loss_scaler), save_path)
    logger.info(f"{save_path} best f1 saved !!!")


def _newest_pth(output_dir):
    if not os.path.isdir(output_dir):
        return None
    cks = [os.path.join(output_dir, d) for d in os.listdir(output_dir) if d.endswith('pth')]
    return max(cks, key=os.path.getmtime) if cks else None


def resume_bestf1_helper(output_dir):
    output_dir = os.path.join(output_dir, 'checkpoint-best-f1')
    os.makedirs(output_dir, exist_ok=True)
    return _newest_pth(output_dir)


def auto_resume_helper(output_dir):
    return _newest_pth(output_dir)


def reduce_tensor(tensor):
    """all-reduce SUM then / world_size (reference :198-202); identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor.clone()
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= dist.get_world_size()
    return rt


class NativeScalerWithGradNormCount:
    """``loss_scaler(loss, optimizer, clip_grad, parameters, update_grad)`` -> grad norm (or None when not updating).
    backward -> [gradient all-reduce] -> global-norm clip -> fused AdamW, all on the stream, no host sync."""
    state_dict_key = "amp_scaler"

    def __init__(self, grad_sync=None):
        self._scale = 1.0
        self.grad_sync = grad_sync          # callable run between backward and the optimizer update (DDP all-reduce)

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        if self.grad_sync is not None:
            self.grad_sync()
        norm = optimizer.clip_grad_norm_(clip_grad if clip_grad is not None else 0.0)
        optimizer.step()
        return norm

    def state_dict(self):
        return {"scale": self._scale, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}

    def load_state_dict(self, state_dict):
        self._scale = float(state_dict.get("scale", 1.0))
