"""world_size-2 gloo tests of the data-parallel path (flat-buffer gradient exchange, parameter broadcast, sampler shards)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mvuld_amd.distributed import GradAllReducer, broadcast_parameters, init_distributed, world_size, get_rank
    r, w, l = init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and world_size() == world and get_rank() == rank
    # DDP construction: rank 0's parameters everywhere
    flat = torch.full((1000,), float(rank + 1))
    broadcast_parameters(flat)
    assert bool((flat == 1.0).all())
    # gradient exchange: two ranges launched "from inside backward", the rest by finish(); result = SUM over ranks
    n = 100_000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradAllReducer(g, max_bucket_elems=30_000)
    red.launch_ranges([(10_000, 45_000), (70_000, 80_000)])
    red.finish()
    want = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    assert torch.equal(g, want), float((g - want).abs().max())
    # second step re-uses the reducer
    g.copy_(torch.ones(n) * (rank + 1))
    red.finish()
    assert bool((g == sum(range(1, world + 1))).all())
    # reduce_tensor = mean over ranks (validate's acc/loss)
    from mvuld_amd.utils_multi import reduce_tensor
    assert float(reduce_tensor(torch.tensor(float(rank)))) == (world - 1) / 2
    # sampler shards: disjoint, equal size, drop_last
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(17))
    s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    s.set_epoch(3)
    idx = torch.tensor(list(s))
    gathered = [torch.zeros_like(idx) for _ in range(world)]
    dist.all_gather(gathered, idx)
    allidx = torch.cat(gathered)
    assert len(idx) == 8 and len(set(allidx.tolist())) == 16
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write("ok")


def test_flat_gradient_allreduce_gloo_world2(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def _worker_wiring(rank, world, port, out):
    """bench.py / main_bigvul.py's data-parallel wiring end to end on a real (CPU) ParamStore under gloo: parameter broadcast,
    attach_gradient_exchange's hooks fired in the order backward fires them, finish(), 1/world factor -- everything but the kernels
    and the backend string."""
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from mvuld_amd import ops
    from mvuld_amd.config import get_config
    from mvuld_amd.distributed import attach_gradient_exchange, broadcast_parameters, gather_cat, init_distributed, world_size
    from mvuld_amd.main_bigvul import build_fused_model
    from mvuld_amd.optimizer import build_optimizer
    init_distributed(backend="gloo")
    cfg = os.path.join(root, "mvuld_amd", "configs", "mySwin", "tiny_plumbing.yaml")
    config = get_config(types.SimpleNamespace(cfg=cfg, opts=["FUSED.DTYPE", "fp32"], batch_size=2, local_rank=0))
    torch.manual_seed(50 + rank)
    model = build_fused_model(config)                    # CPU: parameters and the flat store only, no kernel runs
    build_optimizer(config, model)
    store = model._mv_store
    broadcast_parameters(store.flat)
    ref = [torch.zeros(store.total) for _ in range(world)]
    dist.all_gather(ref, store.flat.detach())
    assert torch.equal(ref[0], ref[1])
    reducer = attach_gradient_exchange(store, max_bucket_elems=1 << 18)
    assert store.grad_scale == 1.0 / world
    launched = []
    orig = reducer.launch_ranges
    reducer.launch_ranges = lambda ranges: (launched.append(list(ranges)), orig(ranges))[1]
    for step in range(2):
        store.grad.copy_(torch.arange(store.total, dtype=torch.float32) % 97 * (rank + 1 + step))
        # the order autograd replays the fused model: graph branch (no tag), Swin stage 3 .. 0, patch embedding ("swin"), text encoder
        for tag in ("swin.layers.3", "swin.layers.2", "swin.layers.1", "swin.layers.0", "swin", "unixcoder"):
            ops.fire_backward_done(tag)
        reducer.finish()
        want = torch.arange(store.total, dtype=torch.float32) % 97 * sum(r + 1 + step for r in range(world))
        assert torch.equal(store.grad, want), float((store.grad - want).abs().max())
    assert len(launched) == 10 and all(len(r) >= 1 for r in launched)          # 5 tagged ranges per step went out from "inside backward"
    covered = sum(b - a for r in launched[:5] for a, b in r)
    enc = sum(p.numel() for n, p in model.named_parameters() if p.requires_grad and n.startswith(("swin.layers.", "unixcoder.")))
    assert enc <= covered <= store.total                                     # every encoder gradient left before finish()
    # validation: per-rank shards differ, the gathered metric inputs do not
    probs = torch.full((3, 2), float(rank))
    allp = gather_cat(probs)
    assert allp.shape == (3 * world, 2) and float(allp.sum()) == 6.0 * sum(range(world))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write("ok")


def _worker_bf16(rank, world, port, out):
    """the bf16-payload exchange (north_star's 464 MB) equals the fp32 exchange to bf16 rounding."""
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mvuld_amd.distributed import GradAllReducer, init_distributed
    init_distributed(backend="gloo")
    n = 300_001
    gen = torch.Generator().manual_seed(7 + rank)
    base = torch.randn(n, generator=gen) * torch.logspace(-6, 2, n)          # eight decades of gradient magnitudes
    got = {}
    for payload in ("fp32", "bf16"):
        g = base.clone()
        red = GradAllReducer(g, max_bucket_elems=1 << 16, payload=payload)
        red.launch_ranges([(1000, 70_000), (200_000, n)])                     # "from inside backward"
        red.finish()                                                          # the rest, then wait (and widen)
        got[payload] = g
    parts = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(parts, base)
    want = sum(parts)
    assert torch.equal(got["fp32"], want)
    # each rank's contribution rounded to bf16 (unit roundoff 2^-8) and one bf16 addition (2^-8 of the sum): |error| <= ~2^-7 (|a| + |b|)
    bound = (parts[0].abs() + parts[1].abs()) * 2.0 ** -7 * 1.01 + 1e-30
    assert bool(((got["bf16"] - want).abs() <= bound).all()), float(((got["bf16"] - want).abs() / bound).max())
    assert float((got["bf16"] - want).abs().max()) > 0.0                      # ... and it really went through bf16
    all_b = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(all_b, got["bf16"])
    assert torch.equal(all_b[0], all_b[1])                                    # every rank ends with the same bits
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(out, "w").write("ok")


def test_bf16_payload_gradient_exchange_gloo_world2(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker_bf16, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_bench_data_parallel_wiring_gloo_world2(tmp_path):
    out = str(tmp_path / "ok")
    mp.spawn(_worker_wiring, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
