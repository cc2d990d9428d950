"""Two ranks on ONE GPU (gloo over the CPU for the exchange): the data-parallel step exactly as bench.py / main_bigvul.py wire it
-- parameter broadcast, per-stage / per-encoder gradient launches fired from inside backward (on the main and the side stream),
finish(), clip with the 1/world factor, fused AdamW -- must leave every rank with identical gradients and parameters."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import types
    import torch.distributed as dist
    from mvuld_amd import ops
    from mvuld_amd.config import get_config
    from mvuld_amd.data import synthetic
    from mvuld_amd.distributed import GradAllReducer, broadcast_parameters, init_distributed, world_size
    from mvuld_amd.main_bigvul import build_fused_model
    from mvuld_amd.models.GraphModel import cross_entropy
    from mvuld_amd.optimizer import build_optimizer
    init_distributed(backend="gloo")
    assert world_size() == world
    dev = torch.device("cuda:0")
    cfg = os.path.join(root, "mvuld_amd", "configs", "mySwin", "tiny_plumbing.yaml")
    config = get_config(types.SimpleNamespace(cfg=cfg, opts=["FUSED.DTYPE", "bf16"], batch_size=2, local_rank=0))
    torch.manual_seed(100 + rank)                       # different initial weights per rank: the broadcast must fix that
    model = build_fused_model(config).to(dev).train()
    opt = build_optimizer(config, model)
    store = model._mv_store
    broadcast_parameters(store.flat)
    store.refresh_working_copy()
    p0 = [torch.zeros(store.total) for _ in range(world)]
    dist.all_gather(p0, store.flat.detach().cpu())
    assert torch.equal(p0[0], p0[1]), "parameters differ right after the broadcast"
    store.grad_scale = 1.0 / world
    reducer = GradAllReducer(store.grad, max_bucket_elems=1 << 20)
    fired = []
    for i in range(4):
        ops.on_backward_done(f"swin.layers.{i}", lambda i=i: (fired.append(f"swin.layers.{i}"), reducer.launch_ranges(store.segment(f"swin.layers.{i}."))))
    ops.on_backward_done("unixcoder", lambda: (fired.append("unixcoder"), reducer.launch_ranges(store.segment("unixcoder."))))
    f = config.FUSED
    g, images, ids, labels = synthetic.make_batch([10 * rank + 1, 10 * rank + 2], config.DATA.IMG_SIZE, f.SEQ_LEN, f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI)
    g, images, ids, labels = g.to(dev), images.to(dev), ids.to(dev), labels.to(dev)
    for step in range(2):
        logits = model(g, images, ids)
        loss, _ = cross_entropy(logits, labels)
        loss.backward()
        reducer.finish()
        torch.cuda.synchronize()
        grads = store.grad.detach().cpu().clone()
        opt.clip_grad_norm_(config.TRAIN.CLIP_GRAD)
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        params = store.flat.detach().cpu().clone()
        both = [torch.zeros_like(params) for _ in range(world)]
        dist.all_gather(both, params)
        bothg = [torch.zeros_like(grads) for _ in range(world)]
        dist.all_gather(bothg, grads)
        if not torch.equal(bothg[0], bothg[1]):
            bad = (bothg[0] != bothg[1]).nonzero().view(-1)
            names = sorted({n for n, _, o, k, _ in store.layout for b in bad[:: max(1, len(bad) // 50)].tolist() if o <= b < o + k})
            raise AssertionError(f"step {step}: {len(bad)} gradient elements differ between ranks, e.g. in {names[:12]}")
        assert float(grads.abs().sum()) > 0 and bool(torch.isfinite(grads).all())
        if not torch.equal(both[0], both[1]):
            bad = (both[0] != both[1]).nonzero().view(-1)
            names = sorted({n for n, _, o, k, _ in store.layout for b in bad[:: max(1, len(bad) // 50)].tolist() if o <= b < o + k})
            raise AssertionError(f"step {step}: {len(bad)} parameter elements differ between ranks (nan: {int(torch.isnan(both[0]).sum())}), "
                                 f"e.g. in {names[:12]}")
        assert bool(torch.isfinite(loss).all())
    assert "unixcoder" in fired and any(t.startswith("swin.layers.") for t in fired)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_two_rank_data_parallel_step_on_one_gpu(tmp_path):
    assert torch.cuda.is_available()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def _worker_rccl(rank, port, out_dir):
    """One rank, backend nccl (= RCCL): every collective of the data-parallel step really goes through RCCL on this GPU -- communicator
    set-up, all-reduces of flat-buffer ranges launched from inside backward (main and side stream), finish(), the broadcast -- and,
    SUM over one rank being the identity, must leave exactly the gradients of the same step without any exchange."""
    os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "MVULD_FORCE_ALLREDUCE": "1"})
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import types
    import torch.distributed as dist
    from mvuld_amd.config import get_config
    from mvuld_amd.data import synthetic
    from mvuld_amd.distributed import attach_gradient_exchange, broadcast_parameters
    from mvuld_amd.main_bigvul import build_fused_model
    from mvuld_amd.models.GraphModel import cross_entropy
    from mvuld_amd.optimizer import build_optimizer
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="env://", world_size=1, rank=0)
    dev = torch.device("cuda:0")
    cfg = os.path.join(root, "mvuld_amd", "configs", "mySwin", "tiny_plumbing.yaml")
    config = get_config(types.SimpleNamespace(cfg=cfg, opts=["FUSED.DTYPE", "bf16"], batch_size=2, local_rank=0))
    f = config.FUSED
    g, images, ids, labels = synthetic.make_batch([3, 4], config.DATA.IMG_SIZE, f.SEQ_LEN, f.TEXT.VOCAB, f.NODES_LO, f.NODES_HI)
    g, images, ids, labels = g.to(dev), images.to(dev), ids.to(dev), labels.to(dev)
    grads = []
    for exchange in (False, True):
        torch.manual_seed(7)
        model = build_fused_model(config).to(dev).eval()      # eval: no dropout / DropPath draws, so the two runs compute the same step
        build_optimizer(config, model)
        store = model._mv_store
        reducer = None
        if exchange:
            t = store.flat.detach().clone()
            broadcast_parameters(store.flat)                 # RCCL broadcast from rank 0 (itself): values unchanged
            assert torch.equal(t, store.flat.detach())
            reducer = attach_gradient_exchange(store, max_bucket_elems=1 << 18)
            assert reducer.force and reducer._active()
            launched = []
            orig = reducer._launch
            reducer._launch = lambda a, b: (launched.append((a, b)), orig(a, b))[1]
        loss, _ = cross_entropy(model(g, images, ids), labels)
        loss.backward()
        if reducer is not None:
            reducer.finish()
            assert len(launched) >= 6 and sum(b - a for a, b in launched) == store.total      # every gradient went through RCCL once
        torch.cuda.synchronize()
        grads.append(store.grad.detach().cpu().clone())
    assert bool(torch.isfinite(grads[0]).all()) and float(grads[0].abs().sum()) > 0
    # atomics make the accumulation order (not the values exchanged) vary from run to run: compare at that noise level
    assert float((grads[0] - grads[1]).norm() / grads[0].norm()) < 1e-4
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, "ok_rccl"), "w").write("ok")


def test_single_rank_exchange_through_rccl(tmp_path):
    mp.spawn(_worker_rccl, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok_rccl")
