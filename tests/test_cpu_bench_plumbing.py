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


def _run_bench(argv, env_extra=None, timeout=240):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("workload", ["sampler", "e2e", "full_tts"])
def test_bench_main_spawns_its_own_ranks(workload):
    """`python bench.py --gpus 2` with NO launcher around it: the parent starts the two ranks itself (a child
    torch.distributed.run; --stub-cpu = gloo ranks with stand-in models), prints exactly one result line, and that line says n_gpus 2,
    names the workload and, for the workloads with a vocoder, shows the waveform was gathered too."""
    p, res = _run_bench(["--gpus", "2", "--stub-cpu", "--steps", "2", "--warmup", "1", "--batch", "3", "--frames", "8", "--workload", workload])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["pipeline"] == workload and res["config"]["utterances_per_gpu"] == 3
    assert res["config"]["gathered"] == (["mel"] if workload == "sampler" else ["mel", "wav"])
    assert abs(res["value"] - 2 * 3 * 8 * 2 / (res["ms_per_step"] * 2e-3)) < 1e-6 * res["value"]


def test_bench_main_refuses_a_world_size_mismatch():
    """a launcher that provides fewer ranks than --gpus asks for must fail loudly, not measure one GPU and label it N"""
    p, res = _run_bench(["--gpus", "2", "--stub-cpu", "--steps", "1", "--warmup", "0", "--batch", "2", "--frames", "8"],
                        env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and res is None
    assert "--gpus 2 but WORLD_SIZE=1" in p.stderr
