"""bench.py's multi-rank plumbing on CPU: two gloo ranks run bench.run_job itself (scatter of the inputs, the timed loop
with its barriers and max-over-ranks reduction, the gather of the mels, and the rank-0-only instrumented step) with a stub
model, so every line of the world > 1 branches executes without a GPU.  A mismatched collective (e.g. a gather inside the
rank-0-only roofline leg) would hang here and fail on the queue timeout."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class _StubModel:
    """mel[b,t,:] = 2 * units[b,t,:80] + spk_id[b]: per-utterance, so the gathered result identifies every shard"""

    def __call__(self, units, volume, spk_id=None, infer=True, infer_speedup=10, method="dpm-solver"):
        assert volume is None and infer and spk_id.dtype == torch.int64 and units.dtype == torch.float32
        return 2.0 * units[..., :80] + spk_id.to(torch.float32)[:, :, None]


class _FakeProfiler:
    def __init__(self):
        self.calls = 0

    def run(self, fn):
        fn()                      # must not contain a collective: only rank 0 gets here
        self.calls += 1
        return [{"name": "conv_dma<BM32 BN64 KT1 S1 U0 BK64 NST2>", "count": 10, "ms": 1.0, "flops": 5e10, "bytes": 1e8},
                {"name": "gn_fused", "count": 4, "ms": 0.5, "flops": 0.0, "bytes": 2e9}]


def _inputs(n, T):
    units = torch.arange(n * T * 1280, dtype=torch.float32).reshape(n, T, 1280) / 1000.0
    spk = (torch.arange(n, dtype=torch.int64) % 7 + 1).reshape(n, 1)
    return units, spk


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "3", "--frames", "8"])
        prof = _FakeProfiler()
        got = {}
        res = bench.run_job(args, rank, world, torch.device("cpu"), _StubModel(), dist=dist, profiler=prof, make_inputs=_inputs,
                            on_output=lambda o: got.setdefault("out", o.clone()))
        ok = True
        if rank == 0:
            units, spk = _inputs(world * 3, 8)
            ok = ok and torch.equal(got["out"], _StubModel()(units, None, spk_id=spk))
            ok = ok and res["n_gpus"] == world and res["steps"] == 2 and res["scaling"] == "weak" and res["value"] > 0
            ok = ok and abs(res["value"] - world * 3 * 8 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]
            ok = ok and prof.calls == 1 and res["roofline"]["kernel"].startswith("conv_dma<BM32") and abs(res["roofline"]["achieved"] - 50.0) < 1e-9
            ok = ok and "gn_fused" in res["roofline"]["hbm_bound_kernels"]
        else:
            ok = res is None and prof.calls == 0
        dist.barrier()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_run_job_multi_rank_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
