#!/usr/bin/env python3
"""Headline benchmark: functions/sec of the fused MVulD train step (SwinV2-base 448^2 + UniXcoder 512 tokens +
~200-node graph head, batch 32 per GPU, bf16) on N MI355X -- BASELINE.json configs[1] (N=1) / configs[2] (N=8).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = forward + cross-entropy + backward + (N>1: flat-buffer RCCL all-reduce) + global-norm clip + fused AdamW +
LR update over one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line (contract in the task prompt)
with two extra objects: ``roofline`` for the dominant kernel family (timed live with HIP events on the launch
stream, in one instrumented step outside the timed region) and ``cpu_baseline`` (the oracle port timed on the host
cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

CFG = os.path.join(ROOT, "mvuld_amd", "configs", "mySwin", "swinv2_base_patch4_window24to28_384to448_1ktoMYDATA_ft.yaml")
PEAK_BF16 = 2.5e15        # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="functions per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--cfg", default=CFG)
    ap.add_argument("--opts", nargs="+", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2, help="functions in the CPU-baseline sample")
    return ap.parse_args()


def build(args, device, rank):
    import types
    from mvuld_amd.config import get_config
    from mvuld_amd.main_bigvul import build_fused_model
    from mvuld_amd.optimizer import build_optimizer
    from mvuld_amd.lr_scheduler import build_scheduler
    from mvuld_amd.data import synthetic
    a = types.SimpleNamespace(cfg=args.cfg, opts=(args.opts or []) + ["FUSED.DTYPE", args.dtype], batch_size=args.batch, local_rank=0)
    config = get_config(a)
    torch.manual_seed(12345)
    model = build_fused_model(config).to(device).train()
    opt = build_optimizer(config, model)
    sched = build_scheduler(config, opt, 1000)
    f = config.FUSED
    idx = [rank * args.batch + i for i in range(args.batch)]
    g, images, ids, labels = synthetic.make_batch(idx, config.DATA.IMG_SIZE, f.SEQ_LEN, f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI)
    g.index()                                    # CSR index on the host, as the data loader's collate does
    lens = (ids != 1).sum(1).to(torch.int32)     # non-pad tokens per function, counted on the host (main_bigvul.model_step_inputs)
    g = g.to(device)
    batch = (g, images.to(device), ids.to(device), labels.to(device), lens)
    return config, model, opt, sched, batch


def algorithmic_flops_per_function(config, n_nodes=200, n_edges=800):
    """forward MACs of the three parts; train step = 3x forward, FLOP = 2 MAC."""
    from mvuld_amd.models.build import build_model
    swin = build_model(config).flops()                                       # counts MACs (reference :645-652)
    t = config.FUSED.TEXT
    L = config.FUSED.SEQ_LEN
    text = t.LAYERS * (4 * t.HIDDEN ** 2 * L + 2 * t.HIDDEN * t.INTERMEDIATE * L + 2 * L * L * t.HIDDEN)
    head = n_nodes * (768 * 2048 + 2048 * 2048 + 2048 * 512 + 8 * 512 * 512) + 100 * (512 * 480 + 4 * 32) \
        + 8 * (4 * 100 * 512 * 512 + 2 * 100 * 100 * 512) + 1024 * 512 + 768 * 512 + 1536 * 2
    return 2.0 * 3.0 * (swin + text + head)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    from mvuld_amd import hip, ops
    hip.LIB.load()
    from mvuld_amd.distributed import GradAllReducer, broadcast_parameters, init_distributed, world_size
    import torch.distributed as dist
    if world > 1:
        init_distributed(local)
    assert world_size() == max(1, world)
    from mvuld_amd.models.GraphModel import cross_entropy

    config, model, opt, sched, batch = build(args, device, rank)
    store = model._mv_store
    broadcast_parameters(store.flat)
    store.refresh_working_copy()
    store.grad_scale = 1.0 / world_size()
    reducer = GradAllReducer(store.grad)
    if world_size() > 1:
        # Swin gradients go out stage by stage, as each stage's backward completes (stage 3 and 2 hold 95 % of them and
        # finish early); patch_embed / norm / whatever else is left goes with finish()
        for _i in range(4):
            ops.on_backward_done(f"swin.layers.{_i}", lambda _i=_i: reducer.launch_ranges(store.segment(f"swin.layers.{_i}.")))
        ops.on_backward_done("unixcoder", lambda: reducer.launch_ranges(store.segment("unixcoder.")))
    g, images, ids, labels, lens = batch
    it = [0]

    def step():
        logits = model(g, images, ids, seq_lens=lens)
        loss, _ = cross_entropy(logits, labels)
        loss.backward()
        reducer.finish()
        opt.clip_grad_norm_(config.TRAIN.CLIP_GRAD)
        opt.step()
        opt.zero_grad()
        sched.step_update(it[0])
        it[0] += 1
        return loss

    def fence():
        torch.cuda.synchronize()
        if world_size() > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3        # host enqueue time per step (the GPU runs behind it)
    fence()
    dt = time.perf_counter() - t0
    if world_size() > 1:
        tt = torch.tensor([dt], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss_val = float(loss)
    n_fn = args.batch * world_size() * args.steps
    value = n_fn / dt
    ms_per_step = dt / args.steps * 1e3

    roofline = None
    if not args.no_kernel_timing:
        hip.TIMING.enable()
        step()
        torch.cuda.synchronize()
        fam = hip.TIMING.summary()
        hip.TIMING.disable()
        roofline = make_roofline(fam, ms_per_step)

    cpu = None
    if rank == 0 and world_size() == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(config, args)

    if rank == 0:
        fl = algorithmic_flops_per_function(config)
        out = {
            "metric": "functions/sec (train step)", "value": round(value, 3), "unit": "functions/s", "n_gpus": world_size(),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "fused MVulD train step: SwinV2-base 448x448 (window 28) + UniXcoder 12x768 @512 tokens + "
                                   "GAT/Rs_GCN head on ~200-node graphs; fwd+CE+bwd+clip+AdamW, random-init weights",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world_size(), "parallelism": f"dp{world_size()}",
                       "algorithmic_tflop_per_step_per_gpu": round(fl * args.batch / 1e12, 2),
                       "achieved_tflops_per_gpu": round(fl * args.batch / (ms_per_step * 1e-3) / 1e12, 2),
                       "frac_of_dense_bf16_peak": round(fl * args.batch / (ms_per_step * 1e-3) / PEAK_BF16, 4),
                       "host_enqueue_ms_per_step": round(host_ms, 2), "final_loss": round(loss_val, 5)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world_size() > 1:
        dist.barrier()
        dist.destroy_process_group()


def make_roofline(fam, ms_per_step):
    """fam: {family: {"ms": total, "n": launches, "flops": .., "bytes": ..}} from the instrumented step."""
    if not fam:
        return None
    name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
    sec = d["ms"] * 1e-3
    top = sorted(((k, round(v["ms"], 3), v["n"]) for k, v in fam.items()), key=lambda x: -x[1])[:8]
    if d["flops"] > 0:
        ach = d["flops"] / sec / 1e12
        r = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_BF16, 4)}
    else:
        ach = d["bytes"] / sec / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": round(ach * 1e9 / PEAK_HBM, 4)}
    r.update({"traffic": measured_traffic(name), "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_hbm_traffic.csv)",
              "kernel": name, "launches_per_step": d["n"], "avg_launch_us": round(d["ms"] * 1e3 / max(1, d["n"]), 2),
              "kernel_ms_per_step": round(d["ms"], 3), "step_ms": round(ms_per_step, 3), "top_kernels_ms": top})
    return r


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command (separate --pmc passes,
    FETCH_SIZE doubled per the gfx950 correction; tools/summarize_profile.py).  None when the summary has no such row."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.csv")
    if not os.path.exists(path):
        return None
    import csv
    for row in csv.DictReader(open(path)):
        if row["kernel"].strip('"') == kernel.split("(")[0]:
            return round((float(row["read_MB_per_launch(x2 corrected)"]) + float(row["write_MB_per_launch"])) * 1e6)
    return None


def cpu_baseline(config, args):
    """The oracle port (plain PyTorch fp32) on the host cores: fused forward + CE + backward + AdamW on `cpu_sample`
    functions of the same workload.  A reported baseline, not the target."""
    from oracle import fused_ref, swin_ref, roberta_ref
    from mvuld_amd import synth
    from mvuld_amd.data import synthetic
    n = args.cpu_sample
    sw = config.MODEL.SWINV2
    scfg = swin_ref.SwinCfg(img_size=config.DATA.IMG_SIZE, embed_dim=sw.EMBED_DIM, depths=list(sw.DEPTHS), num_heads=list(sw.NUM_HEADS),
                            window_size=sw.WINDOW_SIZE, pretrained_window_sizes=list(sw.PRETRAINED_WINDOW_SIZES))
    t = config.FUSED.TEXT
    rcfg = roberta_ref.RobertaCfg(vocab_size=t.VOCAB, hidden_size=t.HIDDEN, num_layers=t.LAYERS, num_heads=t.HEADS,
                                  intermediate_size=t.INTERMEDIATE, max_position=t.MAX_POS)
    shapes = fused_ref.fused_param_shapes(scfg, rcfg)
    sd = {k: synth.synth_param(k, s) for k, s in shapes.items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    g, images, ids, labels = synthetic.make_batch(list(range(n)), config.DATA.IMG_SIZE, config.FUSED.SEQ_LEN, t.VOCAB,
                                                  config.FUSED.NODES_LO, config.FUSED.NODES_HI)
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=0.005)
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    loss, _ = fused_ref.fused_loss(sd, images, ids, g.src, g.dst, g.batch_num_nodes(), g.ndata["_UNIX_NODE_EMB"], g.ndata["pos_emb"],
                                   labels, scfg, rcfg, training=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 4), "unit": "functions/s", "cores": cores, "kind": "port",
            "sample": f"{n} functions, one fused fwd+CE+bwd+clip+AdamW step of the oracle (PyTorch fp32 CPU), {dt:.1f} s"}


if __name__ == "__main__":
    main()
