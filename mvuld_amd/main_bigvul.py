"""MVulD train / validate driver on MI355X -- the reference's ``mvuld/main_bigvul.py`` CLI and step semantics.

Same flags (main_bigvul.py:68-111: ``--seed --cfg --opts --patience --test --batch-size --data-path --test_data_path
--zip --cache-mode --pretrained --resume --myresume --accumulation-steps --use-checkpoint --disable_amp
--amp-opt-level --output --tag --eval --throughput --ngpu --local_rank``), same yaml schema, same launch line
(``python -m torch.distributed.launch --nproc_per_node N main_bigvul.py --cfg <yaml> --batch-size B [--test 1]``;
``--local_rank`` optional, ``LOCAL_RANK/RANK/WORLD_SIZE`` honoured), same step order (:308-345), early stop on
val-F1 (:242-268), metrics (:445-500), LR linear scaling (:545-558) and checkpoint layout.

Differences, all forced by the platform or the north_star:
* the step trains the FUSED model (SwinV2 + UniXcoder + head in one forward/backward) unless ``FUSED.ENABLE False``
  (then it is the reference's head-only step on cached-feature-shaped inputs);
* data is synthetic (``FUSED.SYNTHETIC``; the dataset is not on the box);
* DDP is the flat-buffer RCCL all-reduce of ``distributed.py`` (gloo when no GPU), not torch DDP;
* no ``CUDA_LAUNCH_BLOCKING`` and no per-step ``synchronize()`` (:9,:345): the step never syncs with the host except
  for the ``loss.item()`` the meters need on ``PRINT_FREQ`` steps.
"""
import argparse
import datetime
import gc
import json
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))

from mvuld_amd.config import get_config                                   # noqa: E402
from mvuld_amd.data.bigvul_dataset import bigvul_loader_graph              # noqa: E402
from mvuld_amd.distributed import (GradAllReducer, attach_gradient_exchange, barrier, broadcast_parameters, get_rank, init_distributed,  # noqa: E402
                                   world_size)
from mvuld_amd.logger import create_logger                                 # noqa: E402
from mvuld_amd.lr_scheduler import build_scheduler                         # noqa: E402
from mvuld_amd.metrics import AverageMeter, accuracy, average_precision, binary_prf  # noqa: E402
from mvuld_amd.optimizer import build_optimizer                            # noqa: E402
from mvuld_amd.utils_multi import (NativeScalerWithGradNormCount, auto_resume_helper, load_checkpoint, reduce_tensor,  # noqa: E402
                                   resume_bestf1_helper, save_bestf1_checkpoint)

logger = None


def parse_option(argv=None):
    parser = argparse.ArgumentParser('MVulD fused training and evaluation script', add_help=False)
    parser.add_argument('--seed', type=int, default=12345, help="random seed for initialization")
    parser.add_argument('--cfg', type=str, required=True, metavar="FILE", help='path to config file')
    parser.add_argument("--opts", help="Modify config options by adding 'KEY VALUE' pairs. ", default=None, nargs='+')
    parser.add_argument("--patience", default=50, type=int)
    parser.add_argument('--test', type=int, default=0, help='Train mode=0;Test mode=1')
    parser.add_argument('--batch-size', type=int, help="batch size for single GPU")
    parser.add_argument('--data-path', type=str, help='path to dataset')
    parser.add_argument('--test_data_path', type=str, help='path to test dataset')
    parser.add_argument('--zip', action='store_true', help='use zipped dataset instead of folder dataset')
    parser.add_argument('--cache-mode', type=str, default='part', choices=['no', 'full', 'part'])
    parser.add_argument('--pretrained', help='pretrained weight from checkpoint')
    parser.add_argument('--resume', help='resume from swin checkpoint')
    parser.add_argument('--myresume', help='resume from multimodel checkpoint')
    parser.add_argument('--accumulation-steps', type=int, help="gradient accumulation steps")
    parser.add_argument('--use-checkpoint', action='store_true', help="whether to use gradient checkpointing to save memory")
    parser.add_argument('--disable_amp', action='store_true', help='Disable pytorch amp')
    parser.add_argument('--amp-opt-level', type=str, choices=['O0', 'O1', 'O2'])
    parser.add_argument('--output', default='my_output', type=str, metavar='PATH')
    parser.add_argument('--tag', help='tag of experiment')
    parser.add_argument('--eval', action='store_true', help='Perform evaluation only')
    parser.add_argument('--throughput', action='store_true', help='Test throughput only')
    parser.add_argument('--ngpu', type=int, default=1, help='0 = CPU, 1 = CUDA, 1 < DataParallel')
    parser.add_argument("--local_rank", "--local-rank", type=int, default=None, help='local rank (optional: LOCAL_RANK env is honoured)')
    parser.add_argument('--max-steps', type=int, default=0, help='stop each epoch after this many steps (0 = all; smoke runs)')
    args, unparsed = parser.parse_known_args(argv)
    config = get_config(args)
    return args, config


def act_dtype_of(config):
    return torch.float32 if str(config.FUSED.DTYPE).lower() in ("fp32", "float32", "f32") else torch.bfloat16


def build_fused_model(config):
    from mvuld_amd.models.fused import FusedMVulD
    from mvuld_amd.models.GraphModel import head_class
    from mvuld_amd.models.unixcoder import RobertaConfigLite
    ad = act_dtype_of(config)
    # FUSED.DTYPE fp8: bf16 activations and backward, the encoders' forward QKV / FFN products on the fp8 matrix cores
    from mvuld_amd import ops
    ops.FP8_FWD[0] = str(config.FUSED.DTYPE).lower() in ("fp8", "e4m3")
    if config.FUSED.ENABLE:
        t = config.FUSED.TEXT
        attn_drop = float(t.ATTN_DROPOUT)
        if attn_drop > 0.0 and ad != torch.bfloat16:
            # dropout on attention probabilities lives inside the matrix-core attention kernels (bf16); the fp32 parity mode runs the
            # VALU kernels and is compared with a dropout-free oracle anyway
            print(f"FUSED.DTYPE fp32: FUSED.TEXT.ATTN_DROPOUT {attn_drop} -> 0 (not available in the fp32 parity mode)")
            attn_drop = 0.0
        rc = RobertaConfigLite(vocab_size=t.VOCAB, hidden_size=t.HIDDEN, num_hidden_layers=t.LAYERS, num_attention_heads=t.HEADS,
                               intermediate_size=t.INTERMEDIATE, max_position_embeddings=t.MAX_POS,
                               hidden_dropout_prob=float(t.HIDDEN_DROPOUT), attention_probs_dropout_prob=attn_drop)
        return FusedMVulD(config, rc, ad)
    head = head_class(str(config.FUSED.HEAD))(config=config, act_dtype=ad)
    for n, p in head.named_parameters():          # constructed by the reference, never used in forward: kept in the state dict, not trained
        if n.startswith(tuple(head.unused_parameter_prefixes)):
            p.requires_grad_(False)
    return head


def model_step_inputs(batch, device, pad_token_id=1, image_size=None):
    """Host batch -> device inputs of one step: (g, a, b, target, kwargs).  For the fused model (`b` = token ids) the per-function
    count of non-pad tokens is taken here, while the ids are still on the host: the text encoder then runs pad-free
    (models/unixcoder.py: encode_packed) without a device -> host round trip.  The graph's CSR index is built on the device from the
    edge lists (no host sort, no device -> host copy)."""
    g, a, b, target = batch
    kw = {}
    if b.dtype == torch.int64 and b.dim() == 2 and not b.is_cuda:
        kw["seq_lens"] = (b != pad_token_id).sum(1).to(torch.int32)
    if "_token_ids" in g.ndata and "_UNIX_NODE_EMB" not in g.ndata and b.dtype == torch.int64:
        # per-line token ids from the dataset (data_list.py:240-262 caches them as ndata["_token_ids"]): the node features are computed
        # on the device by the fused model's text encoder (FusedMVulD.forward(node_ids=...)); lengths counted on the host
        nid = g.ndata.pop("_token_ids")
        kw["node_lens"] = (nid != pad_token_id).sum(1).to(torch.int32)
        kw["node_ids"] = nid.to(device, non_blocking=True)
    g = g.to(device)
    g.index()                   # on the device (mvuld_graph_csr_build) unless the loader already built it on the host
    if isinstance(a, (list, tuple)):
        a = device_image_transform(a, device, image_size)
    else:
        a = a.to(device, non_blocking=True)
    return g, a, b.to(device, non_blocking=True), target.to(device, non_blocking=True), kw


def _record_on(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_on(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_on(v, stream)
    elif hasattr(obj, "ndata"):                      # BatchedGraph
        _record_on([obj.src, obj.dst, obj.ndata, obj.edata, obj._index or {}], stream)


def device_batches(loader, device, pad_token_id=1, image_size=None):
    """Iterate `loader` one batch AHEAD of the consumer: while step i computes, batch i+1 goes through model_step_inputs (H2D copies,
    device CSR index, device image transform) on a copy stream; the consumer's stream only waits for an event.  Yields what
    model_step_inputs returns.  (train_one_epoch :308-313 copies every batch synchronously at the top of the step.)"""
    if not (torch.cuda.is_available() and torch.device(device).type == "cuda"):
        for batch in loader:
            yield model_step_inputs(batch, device, pad_token_id, image_size)
        return
    copy = torch.cuda.Stream(device=device)

    def load(batch):
        with torch.cuda.stream(copy):
            out = model_step_inputs(batch, device, pad_token_id, image_size)
            ev = torch.cuda.Event()
            ev.record(copy)
        return out, ev
    it = iter(loader)
    try:
        nxt = load(next(it))
    except StopIteration:
        return
    while nxt is not None:
        cur, ev = nxt
        main = torch.cuda.current_stream(device)
        main.wait_event(ev)
        _record_on(cur, main)
        try:
            nxt = load(next(it))                                    # issued before the consumer enqueues step i: overlaps with it
        except StopIteration:
            nxt = None
        yield cur


_IMAGE_TF = {}


def device_image_transform(images_u8, device, size):
    """Decoded RGB images (uint8 [H, W, 3], any sizes, host or device) -> [B, 3, size, size] float32 on the device: the reference's
    evaluation transform (data/build.py:146-168: PIL bicubic resize, ToTensor, Normalize) run by the HIP kernels of
    data/image_ingest.py on the raw bytes -- the loader thread only decodes, and the H2D copy moves bytes instead of floats."""
    from mvuld_amd.data.image_ingest import DeviceImageTransform
    assert size, "model_step_inputs(..., image_size=config.DATA.IMG_SIZE) is needed for uint8 images"
    tf = _IMAGE_TF.get(size)
    if tf is None:
        tf = _IMAGE_TF[size] = DeviceImageTransform(size)
    outs = [tf(im.to(device, non_blocking=True).contiguous()) for im in images_u8]       # sizes differ per function: one launch pair each
    return torch.cat(outs, 0)


def myMain(config, args, device):
    from mvuld_amd.models.GraphModel import cross_entropy  # noqa: F401
    train_data, val_data, test_data, data_loader_train, data_loader_val, data_loader_test, mixup_fn = bigvul_loader_graph(config)
    print(f"train={len(train_data)} val={len(val_data)} test={len(test_data)}")
    model = build_fused_model(config)
    model.to(device)
    model_without_ddp = model
    optimizer = build_optimizer(config, model)
    store = model._mv_store
    broadcast_parameters(store.flat)                     # DDP construction: rank 0's parameters everywhere
    store.refresh_working_copy()
    if config.FUSED.ENABLE and config.TRAIN.ACCUMULATION_STEPS == 1:
        reducer = attach_gradient_exchange(store)        # exchange launched range by range from inside backward
    else:
        reducer = GradAllReducer(store.grad)             # head-only model / gradient accumulation: one exchange after backward
        store.grad_scale = 1.0 / world_size()
    loss_scaler = NativeScalerWithGradNormCount(grad_sync=reducer.finish)
    n_iter = len(data_loader_train) // max(1, config.TRAIN.ACCUMULATION_STEPS)
    lr_scheduler = build_scheduler(config, optimizer, max(1, n_iter))
    max_accuracy, best_f1 = 0.0, 0.0

    if config.TRAIN.BEST_RESUME:
        os.makedirs(config.MULTI_OUTPUT, exist_ok=True)
        resume_file = resume_bestf1_helper(config.MULTI_OUTPUT)
        if resume_file:
            config.defrost(); config.MODEL.MULTI.RESUME = resume_file; config.freeze()
            logger.info(f'best-f1 resuming from {resume_file}')
        else:
            logger.info(f'no checkpoint found in {config.MULTI_OUTPUT}, ignoring auto resume')
    if config.TRAIN.AUTO_RESUME:
        os.makedirs(config.MULTI_OUTPUT, exist_ok=True)
        resume_file = auto_resume_helper(config.MULTI_OUTPUT)
        if resume_file:
            config.defrost(); config.MODEL.MULTI.RESUME = resume_file; config.freeze()
            logger.info(f'auto resuming from {resume_file}')
    if config.MODEL.MULTI.RESUME:
        max_accuracy, epoch = load_checkpoint(config, model_without_ddp, optimizer, lr_scheduler, loss_scaler, logger)

    if args.test == 0:
        logger.info("----------- Start training -----------")
        start_time = time.time()
        not_f1_inc_cnt = 0
        for epoch in range(config.TRAIN.START_EPOCH, config.TRAIN.EPOCHS):
            data_loader_train.sampler.set_epoch(epoch)
            train_one_epoch(config, model, None, data_loader_train, optimizer, epoch, mixup_fn, lr_scheduler, loss_scaler, device,
                            max_steps=args.max_steps)
            acc1, loss, f1, prauc = validate(config, data_loader_val, model, device)
            if f1 > best_f1 and prauc != 0:
                not_f1_inc_cnt = 0
                logger.info("  Best f1: %s", round(f1, 4))
                logger.info("  prauc: %s", round(prauc, 4))
                logger.info("[%d] Best f1 changed into %.4f\n" % (epoch, round(f1, 4)))
                best_f1 = f1
                output_dir = os.path.join(config.MULTI_OUTPUT, 'checkpoint-best-f1')
                os.makedirs(output_dir, exist_ok=True)
                max_accuracy = max(max_accuracy, acc1)
                if get_rank() == 0:            # one writer; the others wait so that nobody races ahead of a half-written file
                    save_bestf1_checkpoint(config, epoch, model_without_ddp, max_accuracy, optimizer, lr_scheduler, loss_scaler, logger)
                    torch.save(model_without_ddp.state_dict(), os.path.join(output_dir, "pytorch_model.bin"))
                barrier()
            else:
                not_f1_inc_cnt += 1
                logger.info("f1 does not increase for %d epochs", not_f1_inc_cnt)
                if not_f1_inc_cnt > args.patience:
                    logger.info("[%d] Early stop as not_f1_inc_cnt=%d\n" % (epoch, not_f1_inc_cnt))
                    break
            logger.info(f"Accuracy of the network on the {len(val_data)} test images: {acc1:.4f}%")
            logger.info(f"loss of the network on the {len(val_data)} test images: {loss:.4f}%")
            logger.info(f'Max accuracy: {max_accuracy:.4f}%')
            best_f1 = max(best_f1, f1)
            logger.info(f'best f1 : {best_f1:.4f}%')
            gc.collect()
        logger.info('Training time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    else:
        acc1, loss, f1, prauc = validate(config, data_loader_test, model, device)
        logger.info(f"Accuracy of the network on the {len(test_data)} test images: {acc1:.1f}%")
    return model


def train_one_epoch(config, model, criterion, data_loader, optimizer, epoch, mixup_fn, lr_scheduler, loss_scaler, device, max_steps=0):
    from mvuld_amd.models.GraphModel import cross_entropy
    model.train()
    optimizer.zero_grad()
    num_steps = len(data_loader)
    batch_time, loss_meter, norm_meter, scaler_meter = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    start = end = time.time()
    acc = max(1, config.TRAIN.ACCUMULATION_STEPS)
    for idx, (g, a, b, targets, kw) in enumerate(device_batches(data_loader, device, image_size=config.DATA.IMG_SIZE)):
        outputs = model(g, a, b, **kw)
        # CrossEntropyLoss (:298) divided by the accumulation steps (:333); probs = softmax (:330)
        loss, probs = cross_entropy(outputs, targets, loss_scale=1.0 / acc)
        update = (idx + 1) % acc == 0
        grad_norm = loss_scaler(loss, optimizer, clip_grad=config.TRAIN.CLIP_GRAD, parameters=None, update_grad=update)
        if update:
            optimizer.zero_grad()
            lr_scheduler.step_update((epoch * num_steps + idx) // acc)
        if idx % config.PRINT_FREQ == 0:
            loss_meter.update(loss.item(), targets.size(0))
            if grad_norm is not None:
                norm_meter.update(float(grad_norm))
            scaler_meter.update(loss_scaler.state_dict()["scale"])
        batch_time.update(time.time() - end)
        end = time.time()
        if idx % config.PRINT_FREQ == 0:
            lr = optimizer.param_groups[0]['lr']
            wd = optimizer.param_groups[0]['weight_decay']
            memory_used = torch.cuda.max_memory_allocated() / (1024.0 * 1024.0) if torch.cuda.is_available() else 0
            etas = batch_time.avg * (num_steps - idx)
            logger.info(
                f'Train: [{epoch}/{config.TRAIN.EPOCHS}][{idx}/{num_steps}]\t'
                f'eta {datetime.timedelta(seconds=int(etas))} lr {lr:.6f}\t wd {wd:.4f}\t'
                f'time {batch_time.val:.4f} ({batch_time.avg:.4f})\t'
                f'loss {loss_meter.val:.4f} ({loss_meter.avg:.4f})\t'
                f'grad_norm {norm_meter.val:.4f} ({norm_meter.avg:.4f})\t'
                f'loss_scale {scaler_meter.val:.4f} ({scaler_meter.avg:.4f})\t'
                f'mem {memory_used:.0f}MB')
        if max_steps and idx + 1 >= max_steps:
            break
    logger.info(f"EPOCH {epoch} training takes {datetime.timedelta(seconds=int(time.time() - start))}")


@torch.no_grad()
def validate(config, data_loader, model, device):
    from mvuld_amd.models.GraphModel import cross_entropy
    model.eval()
    batch_time, loss_meter, acc1_meter = AverageMeter(), AverageMeter(), AverageMeter()
    outs, probs_all, targets_all = [], [], []
    end = time.time()
    for idx, (g, a, b, targets, kw) in enumerate(device_batches(data_loader, device, image_size=config.DATA.IMG_SIZE)):
        outputs = model(g, a, b, **kw)
        loss, probs = cross_entropy(outputs, targets)
        outs.append(outputs.float()); probs_all.append(probs.float()); targets_all.append(targets.float())
        acc1, _ = accuracy(outputs, targets, topk=(1, 2))
        acc1 = reduce_tensor(acc1)
        loss = reduce_tensor(loss)
        loss_meter.update(loss.item(), targets.size(0))
        acc1_meter.update(acc1.item(), targets.size(0))
        batch_time.update(time.time() - end)
        end = time.time()
        if idx % config.PRINT_FREQ == 0:
            logger.info(f'Test: [{idx}/{len(data_loader)}]\tTime {batch_time.val:.3f} ({batch_time.avg:.3f})\t'
                        f'Loss {loss_meter.val:.4f} ({loss_meter.avg:.4f})\tAcc@1 {acc1_meter.val:.3f} ({acc1_meter.avg:.3f})')
    # every rank evaluates its DistributedSampler shard; P / R / F1 / PR-AUC are taken over ALL shards so that every rank reaches the
    # same "best f1" / early-stop decision (ranks that disagree would leave the others blocked in the next gradient all-reduce)
    from mvuld_amd.distributed import gather_cat
    all_prob = gather_cat(torch.cat(probs_all, 0)).cpu().numpy()
    all_target = gather_cat(torch.cat(targets_all, 0)).cpu().numpy()
    all_predict = all_prob[:, 1] > 0.5                                   # probability threshold (:447)
    P, R, F1Score, TP, FN = binary_prf(all_target, all_predict)
    logger.info(f' * TP {TP:.3f} and (TP+FN) {TP + FN}')
    prauc = 0.0
    if np.isfinite(all_prob[:, 1]).all():
        prauc = average_precision(all_target, all_prob[:, 1])
    acc = float((all_predict == (all_target == 1)).mean())
    logger.info(f' * Acc {acc:.3f} PRECISION {P:.3f} RECALL {R:.3f} F1 {F1Score:.3f} PRAUC {prauc:.3f}')
    return acc1_meter.avg, loss_meter.avg, F1Score, prauc


def main(argv=None):
    global logger
    args, config = parse_option(argv)
    use_cuda = args.ngpu > 0 and torch.cuda.is_available()
    rank, world, local = init_distributed(config.LOCAL_RANK)
    device = torch.device(f"cuda:{local}" if use_cuda else "cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    seed = args.seed                                   # the reference overwrites SEED+rank with args.seed (:533-541)
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
    from mvuld_amd import ops as _ops
    _ops.seed_rng(seed, rank)                          # dropout / DropPath mask streams: per seed AND per rank
    if use_cuda:
        _ops.use_priority_main_stream(device)          # the step's critical chain (image encoder) ahead of the side streams' queues
    ws = world_size()
    # linear LR scaling with the global batch (:545-558)
    scale = config.DATA.BATCH_SIZE * ws / 512.0
    if config.TRAIN.ACCUMULATION_STEPS > 1:
        scale *= config.TRAIN.ACCUMULATION_STEPS
    config.defrost()
    config.TRAIN.BASE_LR = config.TRAIN.BASE_LR * scale
    config.TRAIN.WARMUP_LR = config.TRAIN.WARMUP_LR * scale
    config.TRAIN.MIN_LR = config.TRAIN.MIN_LR * scale
    config.freeze()
    os.makedirs(config.OUTPUT, exist_ok=True)
    logger = create_logger(output_dir=config.OUTPUT, dist_rank=rank, name=f"{config.MODEL.NAME}")
    if rank == 0:
        path = os.path.join(config.OUTPUT, "config.json")
        with open(path, "w") as f:
            f.write(config.dump())
        logger.info(f"Full config saved to {path}")
    logger.info(json.dumps(vars(args)))
    return myMain(config, args, device)


if __name__ == '__main__':
    main()
