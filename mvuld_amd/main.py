"""Stand-alone SwinV2 fine-tune / evaluate / throughput driver -- the reference's ``mvuld/main.py`` (image -> 2 classes; its best-F1
checkpoint is the Swin the fused model starts from, README.md:63-66) on the MI355X kernels.

Same flags as main.py:55-98 (``--cfg --opts --patience --test --batch-size --data-path --zip --cache-mode --pretrained --resume
--myresume --accumulation-steps --use-checkpoint --disable_amp --amp-opt-level --output --tag --eval --throughput --local_rank``),
same step order (:263-283: forward, CrossEntropy / accumulation, clip, AdamW, per-iteration cosine LR), validation metrics
(P / R / F1 / PR-AUC, early stop on F1), ``--throughput`` (:438-455: 50 warm-up + 30 timed forwards) and checkpoint layout.

Mixup / CutMix (``mixup_fn``: :268-269, data/build.py:86-95) run on the device in timm's "batch" mode with host-drawn parameters
(data/mixup.py) and the criterion follows :136-140: SoftTargetCrossEntropy when AUG.MIXUP > 0, LabelSmoothingCrossEntropy when only
MODEL.LABEL_SMOOTHING > 0, CrossEntropyLoss otherwise (one device kernel each).

Differences: the model is this package's SwinTransformerV2 (one fused autograd function per block, bf16 activations instead of
torch.cuda.amp autocast :271); data is synthetic (no dataset on the box), so the image-side augmentations of the training transform
(RandAugment / colour jitter / random erasing, data/build.py:146-168) have nothing to act on and are not built.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))

from mvuld_amd.config import get_config                                   # noqa: E402
from mvuld_amd.distributed import GradAllReducer, barrier, broadcast_parameters, gather_cat, get_rank, init_distributed, world_size  # noqa: E402
from mvuld_amd.logger import create_logger                                 # noqa: E402
from mvuld_amd.lr_scheduler import build_scheduler                         # noqa: E402
from mvuld_amd.metrics import AverageMeter, accuracy, average_precision, binary_prf  # noqa: E402
from mvuld_amd.optimizer import build_optimizer                            # noqa: E402
from mvuld_amd.data.mixup import build_mixup                               # noqa: E402
from mvuld_amd.utils_multi import (NativeScalerWithGradNormCount, auto_resume_helper, load_checkpoint, load_pretrained,  # noqa: E402
                                   reduce_tensor, resume_bestf1_helper, save_bestf1_checkpoint)

logger = None


def parse_option(argv=None):
    parser = argparse.ArgumentParser('Swin Transformer training and evaluation script', add_help=False)
    parser.add_argument('--cfg', type=str, required=True, metavar="FILE", help='path to config file')
    parser.add_argument("--opts", help="Modify config options by adding 'KEY VALUE' pairs. ", default=None, nargs='+')
    parser.add_argument("--patience", default=10, type=int)
    parser.add_argument('--test', type=int, default=0, help='Train mode=0;Test mode=1')
    parser.add_argument('--batch-size', type=int, help="batch size for single GPU")
    parser.add_argument('--data-path', type=str, help='path to dataset')
    parser.add_argument('--zip', action='store_true', help='use zipped dataset instead of folder dataset')
    parser.add_argument('--cache-mode', type=str, default='part', choices=['no', 'full', 'part'])
    parser.add_argument('--pretrained', help='pretrained weight from checkpoint, could be imagenet22k pretrained weight')
    parser.add_argument('--resume', help='resume from checkpoint')
    parser.add_argument('--myresume', help='resume from multimodel checkpoint')
    parser.add_argument('--accumulation-steps', type=int, help="gradient accumulation steps")
    parser.add_argument('--use-checkpoint', action='store_true', help="whether to use gradient checkpointing to save memory")
    parser.add_argument('--disable_amp', action='store_true', help='Disable pytorch amp')
    parser.add_argument('--amp-opt-level', type=str, choices=['O0', 'O1', 'O2'])
    parser.add_argument('--output', default='output', type=str, metavar='PATH')
    parser.add_argument('--tag', help='tag of experiment')
    parser.add_argument('--eval', action='store_true', help='Perform evaluation only')
    parser.add_argument('--throughput', action='store_true', help='Test throughput only')
    parser.add_argument("--local_rank", "--local-rank", type=int, default=None, help='local rank (optional: LOCAL_RANK env is honoured)')
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--max-steps', type=int, default=0, help='stop each epoch after this many steps (0 = all; smoke runs)')
    args, _ = parser.parse_known_args(argv)
    return args, get_config(args)


class SyntheticImages(torch.utils.data.Dataset):
    """(image [3,S,S] f32 ~ post-Normalize statistics, label) by index -- stands in for the rendered CPG images (swin_dataset.py)."""

    def __init__(self, n, base, size):
        self.n, self.base, self.size = n, base, size

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from mvuld_amd.data import synthetic
        return synthetic.make_image(self.base + i, self.size), synthetic.make_label(self.base + i)


def build_loaders(config):
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    f = config.FUSED
    S = config.DATA.IMG_SIZE
    sets = [SyntheticImages(f.SYNTH_TRAIN, 0, S), SyntheticImages(f.SYNTH_VAL, 10_000_000, S), SyntheticImages(f.SYNTH_TEST, 20_000_000, S)]
    ws, rk = world_size(), get_rank()
    samplers = [DistributedSampler(sets[0], num_replicas=ws, rank=rk, shuffle=True, drop_last=True),
                DistributedSampler(sets[1], num_replicas=ws, rank=rk, shuffle=config.TEST.SHUFFLE),
                DistributedSampler(sets[2], num_replicas=ws, rank=rk, shuffle=config.TEST.SHUFFLE)]
    kw = dict(batch_size=config.DATA.BATCH_SIZE, num_workers=0, pin_memory=config.DATA.PIN_MEMORY)
    loaders = [DataLoader(sets[0], sampler=samplers[0], drop_last=True, **kw), DataLoader(sets[1], sampler=samplers[1], **kw),
               DataLoader(sets[2], sampler=samplers[2], **kw)]
    return sets, loaders


def build_criterion(config):
    """main.py:136-140 -> f(outputs, targets, loss_scale) returning (loss, probs)."""
    from mvuld_amd.models.GraphModel import cross_entropy, label_smoothing_cross_entropy, soft_target_cross_entropy
    if config.AUG.MIXUP > 0.0:
        return soft_target_cross_entropy                   # smoothing is handled with the mixup label transform
    if config.MODEL.LABEL_SMOOTHING > 0.0:
        return lambda o, t, loss_scale=1.0: label_smoothing_cross_entropy(o, t, config.MODEL.LABEL_SMOOTHING, loss_scale)
    return cross_entropy


def train_one_epoch(config, model, data_loader, optimizer, epoch, lr_scheduler, loss_scaler, device, max_steps=0, mixup_fn=None, criterion=None):
    from mvuld_amd.models.GraphModel import cross_entropy
    criterion = criterion or cross_entropy
    model.train()
    optimizer.zero_grad()
    num_steps = len(data_loader)
    acc = max(1, config.TRAIN.ACCUMULATION_STEPS)
    batch_time, loss_meter = AverageMeter(), AverageMeter()
    end = time.time()
    for idx, (samples, targets) in enumerate(data_loader):
        samples, targets = samples.to(device, non_blocking=True), targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)                              # :268-269 (soft targets [B, K] from here on)
        outputs = model(samples)
        loss, _ = criterion(outputs, targets, loss_scale=1.0 / acc)                    # criterion / accumulation steps (:272-273)
        update = (idx + 1) % acc == 0
        loss_scaler(loss, optimizer, clip_grad=config.TRAIN.CLIP_GRAD, parameters=None, update_grad=update)
        if update:
            optimizer.zero_grad()
            lr_scheduler.step_update((epoch * num_steps + idx) // acc)
        if idx % config.PRINT_FREQ == 0:
            loss_meter.update(float(loss.detach()), targets.size(0))
            batch_time.update(time.time() - end)
            logger.info(f'Train: [{epoch}/{config.TRAIN.EPOCHS}][{idx}/{num_steps}]\tlr {optimizer.param_groups[0]["lr"]:.6f}\t'
                        f'time {batch_time.val:.4f}\tloss {loss_meter.val:.4f} ({loss_meter.avg:.4f})')
        end = time.time()
        if max_steps and idx + 1 >= max_steps:
            break


@torch.no_grad()
def validate(config, data_loader, model, device):
    from mvuld_amd.models.GraphModel import cross_entropy
    model.eval()
    loss_meter, acc1_meter = AverageMeter(), AverageMeter()
    probs_all, targets_all = [], []
    for samples, targets in data_loader:
        samples, targets = samples.to(device, non_blocking=True), targets.to(device, non_blocking=True)
        outputs = model(samples)
        loss, probs = cross_entropy(outputs, targets)
        acc1, _ = accuracy(outputs, targets, topk=(1, 2))
        loss_meter.update(reduce_tensor(loss).item(), targets.size(0))
        acc1_meter.update(reduce_tensor(acc1).item(), targets.size(0))
        probs_all.append(probs.float())
        targets_all.append(targets.float())
    prob = gather_cat(torch.cat(probs_all, 0)).cpu().numpy()
    tgt = gather_cat(torch.cat(targets_all, 0)).cpu().numpy()
    pred = prob[:, 1] > 0.5
    P, R, F1, TP, FN = binary_prf(tgt, pred)
    prauc = average_precision(tgt, prob[:, 1]) if np.isfinite(prob[:, 1]).all() else 0.0
    logger.info(f' * Acc@1 {acc1_meter.avg:.3f} PRECISION {P:.3f} RECALL {R:.3f} F1 {F1:.3f} PRAUC {prauc:.3f}')
    return acc1_meter.avg, loss_meter.avg, F1, prauc


@torch.no_grad()
def throughput(data_loader, model, device):
    """main.py:438-455: 50 untimed forwards, then 30 timed, images/s of one batch."""
    model.eval()
    for images, _ in data_loader:
        images = images.to(device, non_blocking=True)
        B = images.shape[0]
        for _ in range(50):
            model(images)
        torch.cuda.synchronize()
        logger.info("throughput averaged with 30 times")
        t0 = time.time()
        for _ in range(30):
            model(images)
        torch.cuda.synchronize()
        tput = 30 * B / (time.time() - t0)
        logger.info(f"batch_size {B} throughput {tput}")
        return tput


def my_main(config, args, device):
    from mvuld_amd.main_bigvul import act_dtype_of
    from mvuld_amd.models.build import build_model
    sets, (loader_train, loader_val, loader_test) = build_loaders(config)
    logger.info(f"Creating model:{config.MODEL.TYPE}/{config.MODEL.NAME}")
    model = build_model(config, act_dtype_of(config))
    if config.MODEL.PRETRAINED and not config.MODEL.RESUME:
        load_pretrained(config, model, logger)
    model.to(device)
    logger.info(f"number of params: {sum(p.numel() for p in model.parameters() if p.requires_grad)}")
    if hasattr(model, 'flops'):
        logger.info(f"number of GFLOPs: {model.flops() / 1e9}")
    if config.THROUGHPUT_MODE:
        return throughput(loader_val, model, device)
    optimizer = build_optimizer(config, model)
    store = model._mv_store
    broadcast_parameters(store.flat)
    store.refresh_working_copy()
    reducer = GradAllReducer(store.grad)
    store.grad_scale = 1.0 / world_size()
    loss_scaler = NativeScalerWithGradNormCount(grad_sync=reducer.finish)
    lr_scheduler = build_scheduler(config, optimizer, max(1, len(loader_train) // max(1, config.TRAIN.ACCUMULATION_STEPS)))
    # resume order of main.py:146-181: best-f1 file under OUTPUT (TRAIN.BEST_RESUME) replaces MODEL.RESUME, the checkpoint restores model,
    # optimizer moments, LR schedule, scaler, START_EPOCH and max_accuracy (load_checkpoint); TRAIN.AUTO_RESUME then prefers the newest
    # epoch checkpoint of OUTPUT.
    max_acc = 0.0
    if config.TRAIN.BEST_RESUME:
        f = resume_bestf1_helper(config.OUTPUT)
        if f:
            if config.MODEL.RESUME:
                logger.warning(f"best-resume changing resume file from {config.MODEL.RESUME} to {f}")
            config.defrost(); config.MODEL.RESUME = f; config.freeze()
            logger.info(f'best-f1 resuming from {f}')
        else:
            logger.info(f'no checkpoint found in {config.OUTPUT}/checkpoint-best-f1, ignoring best resume')
    if config.MODEL.RESUME:
        max_acc, _ = load_checkpoint(config, model, optimizer, lr_scheduler, loss_scaler, logger, path=config.MODEL.RESUME)
    if config.TRAIN.AUTO_RESUME:
        f = auto_resume_helper(config.OUTPUT)
        if f:
            if config.MODEL.RESUME:
                logger.warning(f"auto-resume changing resume file from {config.MODEL.RESUME} to {f}")
            config.defrost(); config.MODEL.RESUME = f; config.freeze()
            logger.info(f'auto resuming from {f}')
            max_acc, _ = load_checkpoint(config, model, optimizer, lr_scheduler, loss_scaler, logger, path=f)
        else:
            logger.info(f'no checkpoint found in {config.OUTPUT}, ignoring auto resume')
    if config.EVAL_MODE or args.test:
        return validate(config, loader_test if args.test else loader_val, model, device)
    logger.info("Start training")
    mixup_fn, criterion = build_mixup(config), build_criterion(config)
    best_f1, stale = 0.0, 0
    for epoch in range(config.TRAIN.START_EPOCH, config.TRAIN.EPOCHS):
        loader_train.sampler.set_epoch(epoch)
        train_one_epoch(config, model, loader_train, optimizer, epoch, lr_scheduler, loss_scaler, device, args.max_steps, mixup_fn, criterion)
        acc1, loss, f1, prauc = validate(config, loader_val, model, device)
        max_acc = max(max_acc, acc1)
        if f1 > best_f1 and prauc != 0:
            best_f1, stale = f1, 0
            if get_rank() == 0:
                cfg2 = config.clone(); cfg2.defrost(); cfg2.MULTI_OUTPUT = config.OUTPUT; cfg2.freeze()     # best-f1 checkpoint under OUTPUT (:resume_bestf1_helper)
                save_bestf1_checkpoint(cfg2, epoch, model, max_acc, optimizer, lr_scheduler, loss_scaler, logger)
            barrier()
        else:
            stale += 1
            if stale > args.patience:
                logger.info(f"[{epoch}] Early stop as f1 did not increase for {stale} epochs")
                break
        logger.info(f'Max accuracy: {max_acc:.2f}%')
    return model


def main(argv=None):
    global logger
    args, config = parse_option(argv)
    rank, world, local = init_distributed(config.LOCAL_RANK)
    assert torch.cuda.is_available(), "the Swin fine-tune driver runs on the GPU kernels only"
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    seed = config.SEED + rank if args.seed == 0 else args.seed
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
    from mvuld_amd import ops as _ops
    _ops.seed_rng(seed, rank)                          # dropout / DropPath mask streams: per seed AND per rank
    scale = config.DATA.BATCH_SIZE * world_size() / 512.0 * max(1, config.TRAIN.ACCUMULATION_STEPS)      # linear LR scaling (:main)
    config.defrost()
    config.TRAIN.BASE_LR *= scale; config.TRAIN.WARMUP_LR *= scale; config.TRAIN.MIN_LR *= scale
    config.freeze()
    os.makedirs(config.OUTPUT, exist_ok=True)
    logger = create_logger(output_dir=config.OUTPUT, dist_rank=rank, name=f"{config.MODEL.NAME}")
    if rank == 0:
        with open(os.path.join(config.OUTPUT, "config.json"), "w") as f:
            f.write(config.dump())
    logger.info(json.dumps(vars(args)))
    return my_main(config, args, device)


if __name__ == '__main__':
    main()
