"""The multi-GPU path on CPU: world_size-2 gloo processes exercise the utterance-batch scatter / gather used by
bench.py (RCCL on the GPU box) and check shard-count invariance of the split."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG


def test_shard_range_partitions():
    import sys
    sys.path.insert(0, PKG)
    from lds.shard import shard_range
    for n in (0, 1, 7, 16, 128, 129):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_items, q):
    import sys
    sys.path.insert(0, PKG)
    from lds import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_items * 6, dtype=torch.float32).reshape(n_items, 2, 3) if rank == 0 else torch.empty(0)
        mine = shard.scatter_batch(full, rank, world)
        lo, hi = shard.shard_range(n_items, rank, world)
        ref = torch.arange(n_items * 6, dtype=torch.float32).reshape(n_items, 2, 3)[lo:hi]
        ok = torch.equal(mine, ref)
        out = shard.gather_batch(mine * 2.0, rank, world)        # per-utterance work, then gather on rank 0
        # the same with the shapes known up front (bench.py's path: no object collectives)
        mine2 = shard.scatter_batch(full, rank, world, shape=(n_items, 2, 3))
        sizes = [shard.shard_range(n_items, r, world)[1] - shard.shard_range(n_items, r, world)[0] for r in range(world)]
        out2 = shard.gather_batch(mine2 * 2.0, rank, world, sizes=sizes)
        ok2 = torch.equal(mine2, ref) and ((out2 is None) if rank else torch.equal(out2, out))
        if rank == 0:
            ok = ok and torch.equal(out, torch.arange(n_items * 6, dtype=torch.float32).reshape(n_items, 2, 3) * 2.0)
        else:
            ok = ok and out is None
        ok = ok and ok2
        # integer payloads (speaker ids): dtype announced by the caller or broadcast with the shape
        ids = (torch.arange(n_items, dtype=torch.int64) * 3 + 1).reshape(n_items, 1)
        mine3 = shard.scatter_batch(ids if rank == 0 else torch.empty(0), rank, world, shape=(n_items, 1), dtype=torch.int64)
        mine4 = shard.scatter_batch(ids if rank == 0 else torch.empty(0), rank, world)
        ok = ok and mine3.dtype == torch.int64 and torch.equal(mine3, ids[lo:hi]) and torch.equal(mine4, ids[lo:hi])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 5])
def test_scatter_gather_world2_gloo(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_items
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _bcast_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, PKG)
    from lds import init_weights, shard
    from lds.paramtree import ParamTree
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = {"a.weight": (5, 3), "a.bias": (5,), "b.c.weight": (2, 2, 4)}
        import contextlib
        with (init_weights.deferred() if rank else contextlib.nullcontext()):
            m = ParamTree(shapes, seed=0)                        # rank 0: the seeded values; the others: zeros
        m.register_buffer("steps", torch.tensor([3], dtype=torch.int64))      # non-float entries stay local
        before = {k: v.clone() for k, v in m.state_dict().items()}
        moved = shard.broadcast_state(m.state_dict(), rank, world)
        want = {k: torch.from_numpy(init_weights.init_tensor(k, s, 0)) for k, s in shapes.items()}
        ok = moved == 4 * (15 + 5 + 16)
        ok = ok and all(torch.equal(m.state_dict()[k], want[k]) for k in shapes)
        if rank:
            ok = ok and all(float(before[k].abs().max()) == 0.0 for k in shapes)      # it really was built empty
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_broadcast_state_world3_gloo():
    """ranks >= 1 build their modules empty (init_weights.deferred) and receive rank 0's weights by one broadcast"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 77
    procs = [ctx.Process(target=_bcast_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True), (2, True)]
