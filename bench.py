#!/usr/bin/env python3
"""Headline benchmark: functions/sec of the fused MVulD train step (SwinV2-base 448^2 + UniXcoder 512 tokens +
~200-node graph head, batch 32 per GPU, bf16) on N MI355X -- BASELINE.json configs[1] (N=1) / configs[2] (N=8).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = forward + cross-entropy + backward + (N>1: flat-buffer RCCL all-reduce) + global-norm clip + fused AdamW +
LR update over one synthetic batch already resident in HBM (token ids with their host-side non-pad counts, as the data
loader hands them over: the text encoder runs pad-free).  Rank 0 prints ONE JSON line (contract in the task prompt) with
extra objects: ``roofline`` for the dominant kernel family (timed live with HIP events on the launch stream, in one
instrumented single-stream step outside the timed region), ``cpu_baseline`` (the oracle port timed on the host cores,
BASELINE.md section 4 protocol; N=1 only) and ``inference`` (BASELINE configs[3]: the eval forward at batch 256 captured in a
hipGraph and replayed; N=1 only).  ``--mode infer --batch 256`` prints that leg as the headline line instead.
"""
import argparse
import contextlib
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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8 = BASELINE configs[4]: bf16 step with the encoders' forward QKV / FFN GEMMs in e4m3")
    ap.add_argument("--cfg", default=CFG)
    ap.add_argument("--opts", nargs="+", default=None)
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train: BASELINE configs[1]/[2] (the headline); infer: configs[3], hipGraph-captured eval forward (use --batch 256)")
    ap.add_argument("--text-min-tokens", type=int, default=0,
                    help="non-pad tokens per synthetic function ~ U[this, SEQ_LEN]; 0 (default) = SEQ_LEN: BASELINE's 512-token functions, "
                         "every row full, so the pad-free text encoder skips nothing.  The U[128, 512] case is reported beside it (varlen_text)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-varlen", action="store_true", help="skip the extra U[128, 512]-token leg (varlen_text)")
    ap.add_argument("--no-infer", action="store_true", help="train mode: skip the extra batch-256 hipGraph inference leg")
    ap.add_argument("--graph", action="store_true", help="train mode, N=1: replay the step as ONE captured hipGraph (mvuld_amd/graph_step.py) "
                    "instead of enqueueing it from Python; off by default: on ROCm 7.2 a replay of the ~1900-node, 3-stream graph costs "
                    "~21 us of host time per node and runs ~12 %% slower than the eager step (measured, DESIGN.md)")
    ap.add_argument("--infer-batch", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=4, help="functions per step of the CPU baseline (BASELINE.md section 4: batch 4)")
    ap.add_argument("--cpu-budget-s", type=float, default=240.0, help="safety bound on the CPU baseline leg: the timed runs stop once this much time is spent (the full protocol, 2 warm-ups + 5 runs of ~22 s, takes ~150 s)")
    ap.add_argument("--cpu-threads", type=int, default=64, help="cap on the CPU baseline's torch threads (0 = every physical core)")
    ap.add_argument("--no-fp8", action="store_true", help="train mode: skip the extra fp8 legs (BASELINE configs[4]: train step + batch-256 inference)")
    return ap.parse_args()


def build(args, device, rank):
    import types
    from mvuld_amd.config import get_config
    from mvuld_amd.main_bigvul import build_fused_model
    from mvuld_amd.optimizer import build_optimizer
    from mvuld_amd.lr_scheduler import build_scheduler
    from mvuld_amd.data import synthetic
    a = types.SimpleNamespace(cfg=args.cfg, opts=(args.opts or []) + ["FUSED.DTYPE", args.dtype], batch_size=args.batch, local_rank=0)
    with contextlib.redirect_stdout(sys.stderr):       # the reference-style "=> merge config from ..." line: stdout carries the JSON line only
        config = get_config(a)
    torch.manual_seed(12345)
    model = build_fused_model(config).to(device).train()
    opt = build_optimizer(config, model)
    sched = build_scheduler(config, opt, 1000)
    f = config.FUSED
    idx = [rank * args.batch + i for i in range(args.batch)]
    g, images, ids, labels = synthetic.make_batch(idx, config.DATA.IMG_SIZE, f.SEQ_LEN, f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI,
                                                  tok_lo=args.text_min_tokens or f.SEQ_LEN)
    lens = (ids != 1).sum(1).to(torch.int32)     # non-pad tokens per function, counted on the host (main_bigvul.model_step_inputs)
    g = g.to(device)
    g.index()                                    # CSR index built on the device, as main_bigvul.model_step_inputs does
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
    if args.mode == "train" and not args.graph:
        ops.use_priority_main_stream()         # as main_bigvul.py does: the EAGER step's critical chain on a high-priority stream;
                                               # captured graphs run slower beside a priority stream, so the graph modes stay on the default one
    from mvuld_amd.distributed import attach_gradient_exchange, broadcast_parameters, init_distributed, world_size
    import torch.distributed as dist
    if world > 1:
        init_distributed(local)
    elif os.environ.get("MVULD_FORCE_ALLREDUCE") == "1":
        # rehearsal of the data-parallel control flow on ONE GPU: a single-rank RCCL group, every per-stage all-reduce launched from
        # inside backward (with the stream joins in front of it), finish() -- everything but the wire time
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
    assert world_size() == max(1, world)
    from mvuld_amd.models.GraphModel import cross_entropy

    config, model, opt, sched, batch = build(args, device, rank)
    if args.mode == "infer":
        out = infer_leg(args, config, model, device, world_size(), steps=args.steps, warmup=args.warmup)
        out["cpu_baseline"] = None
        if rank == 0:
            print(json.dumps(out))
        if world_size() > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    store = model._mv_store
    broadcast_parameters(store.flat)
    store.refresh_working_copy()
    reducer = attach_gradient_exchange(store)        # N > 1: per-stage / per-encoder all-reduce launches from inside backward
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

    # --graph (N = 1): the whole step (forward, CE, backward on three streams, clip, AdamW, weight-copy refresh) captured once in a
    # hipGraph and replayed; learning rate / Adam bias corrections / RNG step counter live in device memory (mvuld_amd/graph_step.py).
    # Default: the step is enqueued from Python, kernel by kernel (faster on this ROCm release, see --graph's help).
    graphed, graph_note = None, "eager (kernel-by-kernel enqueue from Python)"
    if world_size() == 1 and args.graph:
        try:
            from mvuld_amd.graph_step import GraphedTrainStep
            from mvuld_amd.models.unixcoder import RobertaModel
            plan = RobertaModel.pack_plan(lens, device, ids.shape[1])      # fixed packing plan of the static batch
            graphed = GraphedTrainStep(model, opt, sched, cross_entropy, (g, images, ids), labels, config.TRAIN.CLIP_GRAD,
                                       model_kwargs={"seq_lens": plan})
            it[0] = graphed.it
            graph_note = "hipGraph replay (captured fwd+CE+bwd+clip+AdamW, 3 streams)"
        except Exception as e:                                     # never lose the headline to a capture problem
            graphed, graph_note = None, f"eager (graph capture failed: {e!r})"
            from mvuld_amd import ops as _ops
            _ops.RNG_OFFSET[0] = None
            opt.dev_hyper = None
            model.max_steps_in_flight = 2
            torch.cuda.synchronize()
    run = (lambda: graphed.step()[0]) if graphed is not None else step

    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3        # host time per step (the GPU runs behind it)
    fence()
    dt = time.perf_counter() - t0
    if world_size() > 1:
        tt = torch.tensor([dt], device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    loss_val = float(loss.detach())
    n_fn = args.batch * world_size() * args.steps
    value = n_fn / dt
    ms_per_step = dt / args.steps * 1e3

    if graphed is not None:
        it[0] = graphed.it
        graphed.close()
    # host cost of enqueueing one step from Python, with the two-steps-in-flight throttle off (3 steps queue up behind the GPU)
    fence()
    model.max_steps_in_flight = 0
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    host_unthrottled_ms = (time.perf_counter() - t0) / 3 * 1e3
    model.max_steps_in_flight = 2
    fence()

    # The same step on functions of U[128, 512] tokens (what the reference's padded [B, 512] rows really hold): the text encoder runs
    # pad-free on the packed tokens, so it does proportionally less work.  Reported beside the headline, never as the headline.
    varlen = None
    if world_size() == 1 and not args.text_min_tokens and not args.no_varlen:
        from mvuld_amd.data import synthetic as _syn
        f_ = config.FUSED
        ids_v = _syn.make_batch(list(range(args.batch)), 8, f_.SEQ_LEN, f_.TEXT.VOCAB, 2, 3, tok_lo=128)[2]
        lens_v = (ids_v != 1).sum(1).to(torch.int32)
        ids_v = ids_v.to(device)
        main_ids, main_lens = ids, lens
        ids, lens = ids_v, lens_v
        for _ in range(max(2, args.warmup)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dtv = time.perf_counter() - t0
        varlen = {"text_tokens_nonpad_frac": round(float(lens_v.sum()) / ids_v.numel(), 4), "value": round(args.batch * args.steps / dtv, 3),
                  "unit": "functions/s", "ms_per_step": round(dtv / args.steps * 1e3, 3)}
        ids, lens = main_ids, main_lens

    # The same step fed from HOST memory, as main_bigvul.py's loop hands batches over (collated batch in pinned memory ->
    # device_batches -> model_step_inputs one batch ahead on a copy stream: asynchronous H2D of 32 x 2.4 MB of images + ids + graph,
    # CSR index built on the device, lengths counted on the host): the PCIe-inclusive rate.  Reported beside the headline; `value` keeps its inputs resident in HBM, as the contract asks.
    host_fed = None
    if world_size() == 1 and not args.no_varlen:
        from mvuld_amd.data import synthetic as _syn
        f_ = config.FUSED
        hb = []
        for k in range(2):
            gh, ih, dh, lh = _syn.make_batch([1000 * (k + 1) + i for i in range(args.batch)], config.DATA.IMG_SIZE, f_.SEQ_LEN, f_.TEXT.VOCAB,
                                             f_.NODES_LO, f_.NODES_HI, tok_lo=f_.SEQ_LEN)
            gh.src, gh.dst = gh.src.pin_memory(), gh.dst.pin_memory()
            gh.ndata = {k: v.pin_memory() for k, v in gh.ndata.items()}
            hb.append((gh, ih.pin_memory(), dh.pin_memory(), lh.pin_memory()))
        keep = (g, images, ids, labels, lens)
        from mvuld_amd.main_bigvul import device_batches

        def host_steps(n):                      # n steps through the prefetching iterator main_bigvul.train_one_epoch uses
            nonlocal g, images, ids, labels, lens
            for g, images, ids, labels, kw in device_batches((hb[k % 2] for k in range(n)), device):
                lens = kw.get("seq_lens")
                step()
        host_steps(max(2, args.warmup))
        fence()
        t0 = time.perf_counter()
        host_steps(args.steps)
        fence()
        dth = time.perf_counter() - t0
        host_fed = {"value": round(args.batch * args.steps / dth, 3), "unit": "functions/s", "ms_per_step": round(dth / args.steps * 1e3, 3),
                    "h2d_MB_per_step": round(sum(t.numel() * t.element_size() for t in hb[0][1:]) / 1e6, 1)}
        g, images, ids, labels, lens = keep

    roofline = None
    if not args.no_kernel_timing:
        hip.TIMING.enable()
        step()
        torch.cuda.synchronize()
        fam = hip.TIMING.summary()
        hip.TIMING.disable()
        roofline = make_roofline(fam, ms_per_step)

    inference = None
    if not args.no_infer and world_size() == 1:
        try:
            inf = infer_leg(args, config, model, device, 1)
            inference = {k: inf[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup")}
            inference.update({"batch_per_gpu": args.infer_batch, "graph_replay_equals_eager": inf["config"]["graph_replay_equals_eager"],
                              "frac_of_dense_bf16_peak": inf["config"]["frac_of_dense_bf16_peak"]})
        except Exception as e:                      # the headline must still print
            inference = {"error": repr(e)}

    cpu = None
    if rank == 0 and world_size() == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(config, args)

    # BASELINE configs[4] beside the headline (N = 1): the same step and the same batch-256 inference with the encoders' forward QKV / FFN
    # products on e4m3 operands -- a second model (FUSED.DTYPE fp8), built after everything bf16 has been measured
    fp8 = None
    host_ms_reported = host_ms if graphed is not None else host_unthrottled_ms
    if rank == 0 and world_size() == 1 and args.dtype == "bf16" and not args.no_fp8 and not args.no_infer:
        try:
            import copy
            a8 = copy.copy(args)
            a8.dtype = "fp8"
            graphed = None                                  # the captured train step and its pools hold the bf16 model alive
            del model, opt, sched, store, reducer
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            config8, model8, opt8, sched8, batch8 = build(a8, device, rank)
            g8, im8, id8, lb8, ln8 = batch8
            model8._mv_store.refresh_working_copy()
            it8 = [0]

            def step8():
                lg = model8(g8, im8, id8, seq_lens=ln8)
                l8, _ = cross_entropy(lg, lb8)
                l8.backward()
                opt8.clip_grad_norm_(config8.TRAIN.CLIP_GRAD)
                opt8.step()
                opt8.zero_grad()
                sched8.step_update(it8[0])
                it8[0] += 1
                return l8
            for _ in range(max(3, args.warmup)):          # the first pass calibrates the quantisation sites
                step8()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                l8 = step8()
            fence()
            d8 = time.perf_counter() - t0
            fp8 = {"train": {"value": round(args.batch * args.steps / d8, 3), "unit": "functions/s", "ms_per_step": round(d8 / args.steps * 1e3, 3),
                             "final_loss": round(float(l8), 5)}}
            inf8 = infer_leg(a8, config8, model8, device, 1)
            fp8["inference"] = {k: inf8[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")}
            fp8["inference"]["batch_per_gpu"] = args.infer_batch
            fp8["what"] = "BASELINE configs[4] on one GPU: forward QKV / FFN GEMMs of both encoders in OCP e4m3 (fp32 accumulate), rest bf16"
        except Exception as e:                          # the headline must still print -- but not silently
            fp8 = {"error": repr(e)}
            print(f"bench.py: fp8 leg failed: {e!r}", file=sys.stderr, flush=True)
        finally:
            from mvuld_amd import ops as _ops8
            _ops8.FP8_FWD[0] = False

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
                       "step_launch": graph_note,
                       "host_enqueue_ms_per_step": round(host_ms_reported, 3),
                       "host_enqueue_eager_ms_per_step": round(host_unthrottled_ms, 2),
                       "text_tokens_nonpad_frac": round(float(lens.sum()) / ids.numel(), 4), "final_loss": round(loss_val, 5)},
            "roofline": roofline, "cpu_baseline": cpu, "inference": inference, "fp8": fp8, "varlen_text": varlen, "host_fed_inputs": host_fed,
        }
        print(json.dumps(out))
    if world_size() > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


def infer_leg(args, config, model, device, world, steps=5, warmup=2):
    """BASELINE configs[3]: inference-only fused forward at `infer_batch` functions per GPU, captured ONCE in a hipGraph (both
    streams, every launch through the C ABI, pad-free text encoder from a precomputed packing plan) and replayed.
    Returns the dict bench.py prints (mode infer) or embeds under "inference" (mode train)."""
    from mvuld_amd import hip
    from mvuld_amd.data import synthetic
    from mvuld_amd.models.unixcoder import RobertaModel
    B = args.infer_batch
    f = config.FUSED
    rank = int(os.environ.get("RANK", 0))
    idx = [10_000 + rank * B + i for i in range(B)]
    g, images, ids, _ = synthetic.make_batch(idx, config.DATA.IMG_SIZE, f.SEQ_LEN, f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI,
                                             tok_lo=args.text_min_tokens or f.SEQ_LEN)      # full 512-token rows, as the headline
    g.index()
    plan = RobertaModel.pack_plan((ids != 1).sum(1), device, ids.shape[1])      # device-resident cu_seqlens: no host work inside the graph
    g, images, ids = g.to(device), images.to(device), ids.to(device)
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            eager = model(g, images, ids, seq_lens=plan).float().clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model(g, images, ids, seq_lens=plan)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = model(g, images, ids, seq_lens=plan)
        for _ in range(warmup):
            graph.replay()
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        same = bool(torch.equal(out.float(), eager))
        roofline = None
        if not args.no_kernel_timing:
            hip.TIMING.enable()
            model(g, images, ids, seq_lens=plan)
            torch.cuda.synchronize()
            fam = hip.TIMING.summary()
            hip.TIMING.disable()
            roofline = make_roofline(fam, dt / steps * 1e3)
        del graph, out
    model.train(was_training)
    fl = algorithmic_flops_per_function(config) / 3.0                           # forward only
    ms = dt / steps * 1e3
    return {"metric": "functions/sec (inference, hipGraph-captured fused forward)", "value": round(B * world * steps / dt, 3), "unit": "functions/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "fused MVulD eval forward (BASELINE configs[3]): SwinV2-base 448x448 + UniXcoder 12x768 @512 tokens (pad-free) + "
                                   "GAT/Rs_GCN head, one hipGraph replay per step", "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": f"replicas x{world}", "graph_replay_equals_eager": same,
                       "algorithmic_tflop_per_step_per_gpu": round(fl * B / 1e12, 2),
                       "frac_of_dense_bf16_peak": round(fl * B / (ms * 1e-3) / PEAK_BF16, 4)},
            "roofline": roofline}


# Kernel families of the instrumented step -> the groups `roofline` ranks.  Every attention launch (forward, the backward passes, the
# matrix-core and the VALU forms) is ONE family, as every NT GEMM variant already is: ranking them split hid the largest time share.
# Algorithmic FLOPs of attention: forward 4 N^2 hd, backward 10 N^2 hd per (window | sequence, head) -- what a kernel recomputes on top
# of that (the text encoder's two-pass backward forms the scores and dP twice) is overhead, not work.
FAMILY_GROUPS = {
    "attention": ("attn_fwd_mfma", "attn_bwd_mfma", "attn_fwd_simple", "attn_bwd_simple"),
    "gemm_nt_mfma_bf16": ("gemm_nt_mfma_bf16", "gemm_nt_mfma_fp8"),
}
FLOP_REPRICE = {}
TRAFFIC_KERNELS = {
    "attention": ("attn_fwd_win_k", "attn_fwd_mfma_k", "attn_bwd_dq_mfma_k", "attn_bwd_dkv_mfma_k", "attn_bwd_dbias_mfma_k", "attn_dbias_reduce_k",
                  "attn_bwd_fused_win_k"),
    "gemm_nt_mfma_bf16": ("gemm_nt_mfma_bf16",),
    "gemm_tn_wgrad": ("gemm_tn_wgrad", "gemm_tn256_group_k", "gemm_tn256_group_reduce_k", "gemm_tn256_reduce_k"),
}


def group_families(fam):
    """{group: {"ms", "n", "flops", "bytes", "members": {family: ms}}}; families outside FAMILY_GROUPS stay groups of their own."""
    owner = {m: grp for grp, members in FAMILY_GROUPS.items() for m in members}
    out = {}
    for name, d in fam.items():
        grp = owner.get(name, name)
        o = out.setdefault(grp, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0, "members": {}})
        o["ms"] += d["ms"]
        o["n"] += d["n"]
        o["flops"] += d["flops"] * FLOP_REPRICE.get(name, 1.0)
        o["bytes"] += d["bytes"]
        o["members"][name] = round(d["ms"], 3)
    return out


def roofline_of(name, d, ms_per_step):
    sec = d["ms"] * 1e-3
    if d["flops"] > 0:
        ach = d["flops"] / sec / 1e12
        r = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": round(ach * 1e12 / PEAK_BF16, 4)}
    else:
        ach = d["bytes"] / sec / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": round(ach * 1e9 / PEAK_HBM, 4)}
    r.update({"traffic": measured_traffic(name, d["n"]),
              "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE over the family's kernels, newest profiles/rNN_hbm_traffic.csv)",
              "kernel": name, "members_ms": d["members"], "launches_per_step": d["n"], "avg_launch_us": round(d["ms"] * 1e3 / max(1, d["n"]), 2),
              "kernel_ms_per_step": round(d["ms"], 3), "algorithmic_tflop_per_step": round(d["flops"] / 1e12, 3), "step_ms": round(ms_per_step, 3)})
    return r


def make_roofline(fam, ms_per_step):
    """fam: {family: {"ms": total, "n": launches, "flops": .., "bytes": ..}} from the instrumented step (HIP events on the launch stream).
    `roofline` = the GROUP with the largest time share (all attention launches are one group, all NT GEMMs another); the runner-up rides
    along as `runner_up` so that the two large families are always both visible."""
    if not fam:
        return None
    groups = group_families(fam)
    ranked = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
    r = roofline_of(ranked[0][0], ranked[0][1], ms_per_step)
    if len(ranked) > 1:
        r["runner_up"] = roofline_of(ranked[1][0], ranked[1][1], ms_per_step)
    r["top_kernels_ms"] = [(k, round(v["ms"], 3), v["n"]) for k, v in ranked[:8]]
    from mvuld_amd import hip
    r["event_bracket_us_subtracted"] = round(hip.TIMING.bracket_us or 0.0, 2)      # per launch: what the event pair itself costs (hip.py: calibrate)
    return r


def measured_traffic(group, launches_per_step):
    """HBM bytes per launch (launch = one C-ABI call of the family, as `achieved` counts them) from the committed PMC summary of this same
    command (separate --pmc passes, FETCH_SIZE doubled per the gfx950 correction; tools/summarize_profile.py): the bytes of every kernel of
    the family over the pass / the pass's step-equivalents / launches per step.  None when the summary has no such rows."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_hbm_traffic.csv") for r in (4, 3, 2)) if os.path.exists(q)), None)
    if path is None or not launches_per_step:
        return None
    import csv
    kernels = TRAFFIC_KERNELS.get(group, (group,))
    total, steps = 0.0, None
    for row in csv.DictReader(open(path)):
        k = row["kernel"].strip('"')
        if k == "adamw_k":
            steps = float(row["launches"]) / 2.0          # two AdamW launches (decay / no-decay group) per step-equivalent of the pass
        if any(m in k for m in kernels):
            total += float(row["launches"]) * (float(row["read_MB_per_launch(x2 corrected)"]) + float(row["write_MB_per_launch"])) * 1e6
    if total == 0.0 or not steps:
        return None
    return round(total / steps / launches_per_step)


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or 0
    except Exception:
        phys = 0
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return model, phys, avail


def cpu_baseline(config, args):
    """the oracle port (plain PyTorch fp32 -- oracle/fused_ref.py; test infrastructure, imported here only as
    the reported baseline) on the GPU box's host cores: one fused train step (forward + CE + backward + clip + AdamW) on a batch of
    `cpu_sample` functions of the same synthetic workload, 2 warm-ups, median of
    5 timed runs (fewer only if the `cpu_budget_s` safety bound runs out), torch threads = physical cores available to this process.  A reported baseline, not the target."""
    from oracle import fused_ref, swin_ref, roberta_ref
    from mvuld_amd import synth
    from mvuld_amd.data import synthetic
    n = args.cpu_sample
    model_name, phys, avail = cpu_info()
    cores = max(1, min(phys, avail) if phys else avail)
    if args.cpu_threads > 0:
        cores = min(cores, args.cpu_threads)
    torch.set_num_threads(cores)
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

    def one():
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, _ = fused_ref.fused_loss(sd, images, ids, g.src, g.dst, g.batch_num_nodes(), g.ndata["_UNIX_NODE_EMB"], g.ndata["pos_emb"],
                                       labels, scfg, rcfg, training=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        return time.perf_counter() - t0

    t_all = time.perf_counter()
    warm = [one(), one()]                                   # 2 warm-ups, median of >= 5 runs
    runs = []
    while len(runs) < 5 and (not runs or time.perf_counter() - t_all < args.cpu_budget_s):
        runs.append(one())
    runs.sort()
    med = runs[len(runs) // 2]
    full = len(runs) >= 5
    return {"value": round(n / med, 4), "unit": "functions/s", "cores": cores, "kind": "port",
            "cpu_model": model_name, "physical_cores": phys, "threads_used": cores, "warmups": len(warm), "runs": len(runs),
            "protocol": ("BASELINE.md section 4 protocol (b): batch-4 fused forward + backward + AdamW step, 2 warm-ups, median of 5 runs"
                         if full else f"BASELINE.md section 4 asks 2 warm-ups and the median of >= 5 runs; --cpu-budget-s {args.cpu_budget_s:g} s "
                                      f"ran out after {len(runs)} timed runs") +
                        f"; threads = min(physical cores, --cpu-threads {args.cpu_threads}): the oracle's batch-4 operators do not scale past one socket",
            "sample": f"batch {n}, fused fwd+CE+bwd+clip+AdamW step of the oracle (PyTorch fp32 CPU): 2 warm-ups ({warm[0]:.1f} s, {warm[1]:.1f} s), "
                      f"median of {len(runs)} timed runs = {med:.2f} s/step"}


if __name__ == "__main__":
    main()
