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
