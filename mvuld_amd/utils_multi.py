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


def load_checkpoint(config, model, optimizer, lr_scheduler, loss_scaler, logger):
    logger.info(f"==============> Resuming form {config.MODEL.MULTI.RESUME}....................")
    if config.MODEL.MULTI.RESUME.startswith('https'):
        raise RuntimeError("no network on this box: pass a local checkpoint path")
    checkpoint = torch.load(config.MODEL.MULTI.RESUME, map_location='cpu', weights_only=False)
    model.load_state_dict(checkpoint['model'], strict=False)
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
        logger.info(f"=> loaded successfully '{config.MODEL.MULTI.RESUME}' (epoch {checkpoint['epoch']})")
        if 'max_accuracy' in checkpoint:
            max_accuracy = checkpoint['max_accuracy']
    del checkpoint
    return max_accuracy, epoch


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
